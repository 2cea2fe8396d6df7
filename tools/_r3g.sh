for rep in 1 2; do for lib in path_tracer_amd/libptmi.so; do
  for cfg in "1" "8" "1 0 cornell_mesh:6 8" "1 0 cornell_mesh:7 8" "1 0 cornell_spheres:4 8" "1 0 cornell_mixed 32"; do PTMI_LIB=$PWD/$lib timeout -k 10 120 python tools/one_frame.py $cfg 2>/dev/null | grep -v "^B" | sed "s|^|$lib |"; done
done; done
timeout -k 10 200 python tools/shard_probe.py 2 4 8
timeout -k 10 120 python tools/frame_bench.py 2>/dev/null | tail -2
