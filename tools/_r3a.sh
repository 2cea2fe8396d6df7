set -e
mkdir -p gpurun_out/r3a
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest.log 2>&1 || { tail -30 gpurun_out/r3a/pytest.log; exit 1; }
tail -3 gpurun_out/r3a/pytest.log
tools/ab_bench.sh path_tracer_amd/libptmi.so build/variants/r2.so > gpurun_out/r3a/ab.log 2>&1
cat gpurun_out/r3a/ab.log
PTMI_LIB=$PWD/build/variants/stats.so timeout -k 10 200 python tools/step_stats2.py cornell_box 16 > gpurun_out/r3a/stats2.md 2>&1
cat gpurun_out/r3a/stats2.md
PTMI_LIB=$PWD/build/variants/stats_r2.so timeout -k 10 200 python tools/step_stats.py cornell_box 16 > gpurun_out/r3a/stats_r2.md 2>&1
tail -4 gpurun_out/r3a/stats_r2.md
