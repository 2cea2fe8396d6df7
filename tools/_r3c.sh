set -e
mkdir -p gpurun_out/r3c
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3c/pytest.log 2>&1 || { tail -30 gpurun_out/r3c/pytest.log; exit 1; }
tail -3 gpurun_out/r3c/pytest.log
tools/ab_bench.sh path_tracer_amd/libptmi.so build/variants/r2.so > gpurun_out/r3c/ab.log 2>&1
cat gpurun_out/r3c/ab.log
PTMI_LIB=$PWD/build/variants/stats3.so timeout -k 10 200 python tools/step_stats2.py cornell_box 16 > gpurun_out/r3c/stats3.md 2>&1
cat gpurun_out/r3c/stats3.md
