set -e
mkdir -p gpurun_out/r3e
PTMI_LIB=$PWD/build/variants/fused1.so timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3e/pytest_fused1.log 2>&1 || { tail -30 gpurun_out/r3e/pytest_fused1.log; exit 1; }
tail -2 gpurun_out/r3e/pytest_fused1.log
for rep in 1 2; do for lib in build/variants/fused0.so build/variants/fused1.so; do
  for cfg in "1" "8" "4" "2"; do PTMI_LIB=$PWD/$lib timeout -k 10 120 python tools/one_frame.py $cfg 2>/dev/null | grep -v "^B" | sed "s|^|$lib |"; done
done; done
for lib in build/variants/fused0.so build/variants/fused1.so; do PTMI_LIB=$PWD/$lib timeout -k 10 120 python tools/frame_bench.py 2>/dev/null | tail -3 | sed "s|^|$lib |"; done
