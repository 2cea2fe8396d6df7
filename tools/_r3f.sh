set -e
mkdir -p gpurun_out/r3f
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3f/pytest.log 2>&1 || { tail -30 gpurun_out/r3f/pytest.log; exit 1; }
tail -2 gpurun_out/r3f/pytest.log
tools/ab_bench.sh path_tracer_amd/libptmi.so build/variants/r2.so
for rep in 1 2; do for lib in path_tracer_amd/libptmi.so build/variants/r2.so; do
  for cfg in "8" "1 0 cornell_mesh:6 8" "1 0 cornell_mesh:7 8" "1 0 cornell_spheres:4 8" "1 0 cornell_mixed 32"; do PTMI_LIB=$PWD/$lib timeout -k 10 120 python tools/one_frame.py $cfg 2>/dev/null | grep -v "^B" | sed "s|^|$lib |"; done
done; done
