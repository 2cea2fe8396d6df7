# the bench line of every configuration (GPU box); each prints one JSON line into gpurun_out/bench_<ROUND>/<config>.json
ROUND=${ROUND:-r04}
mkdir -p gpurun_out/bench_$ROUND
for c in ${CONFIGS:-cornell mesh82k atrium mesh328k mixed spheres}; do
  echo "== $c $(date +%T)"
  timeout -k 10 900 python bench.py --config $c > gpurun_out/bench_$ROUND/$c.json 2> gpurun_out/bench_$ROUND/$c.err || { echo "bench $c failed"; tail -3 gpurun_out/bench_$ROUND/$c.err; }
  python - $c $ROUND <<'PY'
import json, sys
c, rnd = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(f"gpurun_out/bench_{rnd}/{c}.json").read().strip().split("\n")[-1])
    km = d["kernel_ms"] or {}
    print(c, d["config"]["workload"], "|", round(d["ms_per_step"], 2), "ms/step |", round(d["value"]), "Mray/s traversed,", round(d["config"]["cast_Mray_per_s"]), "cast | kernel_ms", {k: round(v, 1) for k, v in km.items() if k != "source"},
          "| roofline frac", round(d["roofline"]["frac"], 3), "alg B/ray", round(d["roofline"]["algorithmic_bytes_per_ray"]), "counter/alg", d["roofline"].get("counter_over_algorithmic"), "| shade", d.get("roofline_shade", {}).get("counter_GBps"), "| cpu", round(d.get("cpu_baseline", {}).get("value", 0), 1), "| state GiB", round(d["config"]["state_GiB"], 1))
except Exception as e:
    print(c, "unreadable:", e)
PY
done
