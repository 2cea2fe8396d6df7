set -e
mkdir -p gpurun_out/r3d
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "checkpoint or resume" > gpurun_out/r3d/pytest.log 2>&1 || { tail -30 gpurun_out/r3d/pytest.log; exit 1; }
tail -2 gpurun_out/r3d/pytest.log
timeout -k 10 300 python bench.py > gpurun_out/r3d/bench_cornell.json 2> gpurun_out/r3d/bench_cornell.err || { tail gpurun_out/r3d/bench_cornell.err; exit 1; }
timeout -k 10 300 python bench.py --config mesh82k --spp 32 > gpurun_out/r3d/bench_mesh82k.json 2> gpurun_out/r3d/bench_mesh82k.err || { tail gpurun_out/r3d/bench_mesh82k.err; exit 1; }
timeout -k 10 300 python bench.py --config mixed --spp 64 --no-cpu-baseline > gpurun_out/r3d/bench_mixed.json 2> gpurun_out/r3d/bench_mixed.err || { tail gpurun_out/r3d/bench_mixed.err; exit 1; }
python - <<'PY'
import json
for n in ("cornell","mesh82k","mixed"):
    d=json.loads(open(f"gpurun_out/r3d/bench_{n}.json").read().strip().split("\n")[-1])
    print(n, round(d["ms_per_step"],2), "ms", round(d["value"]), "Mray/s traversed", round(d["config"]["cast_Mray_per_s"]), "cast;", d["kernel_ms"] and {k:round(v,2) for k,v in d["kernel_ms"].items() if k!="source"}, "roofline frac", round(d["roofline"]["frac"],3), d["roofline"]["algorithmic_bytes_per_ray"], d.get("cpu_baseline",{}).get("value"))
PY
