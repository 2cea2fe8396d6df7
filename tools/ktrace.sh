#!/bin/bash
# Kernel-trace totals of one 1080p frame for library variants (GPU box): tools/ktrace.sh [one_frame args in quotes] a.so b.so ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ARGS=$1; shift
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  O=$R/gpurun_out/kt/$tag
  rm -rf "$O" && mkdir -p "$O"
  export PTMI_LIB=$R/$lib
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o r1 -- python "$R/tools/one_frame.py" $ARGS > "$O/log.txt" 2>&1 || { tail -5 "$O/log.txt"; exit 1; }
  echo "== $lib ($ARGS)"
  python - "$O" <<'PY'
import sys, glob, re, pandas as pd
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
ks = pd.read_csv(f)
short = lambda n: re.sub(r"\(.*", "", re.sub(r"^void ", "", re.sub(r"pt::\(anonymous namespace\)::", "", n)))
for _, r in ks.iterrows():
    n = short(r["Name"])
    if n.startswith("k_"): print(f"{n:45s} calls {r['Calls']:4d} total {r['TotalDurationNs'] / 1e6:8.3f} ms  avg {r['AverageNs'] / 1e3:8.1f} us")
PY
  find "$O" -name "*.csv" -size +5M -delete
done
