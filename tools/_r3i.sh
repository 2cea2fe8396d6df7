for rep in 1 2; do for lib in build/variants/noshuf.so path_tracer_amd/libptmi.so; do for sc in "cornell_mesh 8 6" "cornell_mesh 8 7"; do
PTMI_LIB=$PWD/$lib timeout -k 10 120 python tools/scene_bench.py $sc 2>/dev/null | sed "s|^|$lib |"; done; done; done
