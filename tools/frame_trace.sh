#!/bin/bash
# Is the interactive 1-spp frame bound by the GPU or by launching?  Kernel-trace of tools/frame_bench.py: summed kernel durations per frame
# against the frame time the same run reports.  (GPU box)   tools/frame_trace.sh [frames]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-200}
O=$R/gpurun_out/frame_trace
rm -rf "$O" && mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o r1 -- python "$R/tools/frame_bench.py" $N > "$O/log.txt" 2>&1 || { tail -5 "$O/log.txt"; exit 1; }
tail -1 "$O/log.txt"
python - "$O" $N <<'PY'
import sys, glob, re, pandas as pd
o, n = sys.argv[1], int(sys.argv[2])
frames = 5 + 2 * n            # warm-up + static + moving
ks = pd.read_csv(glob.glob(o + "/**/*kernel_stats.csv", recursive=True)[0])
tot = ks["TotalDurationNs"].sum() / 1e6
calls = ks["Calls"].sum()
print(f"{calls} launches, {tot:.1f} ms of kernel time over {frames} frames + presents = {tot / frames:.3f} ms and {calls / frames:.1f} launches per frame")
short = lambda s: re.sub(r"\(.*", "", re.sub(r"^void ", "", re.sub(r"pt::\(anonymous namespace\)::", "", s)))
for _, r in ks.head(12).iterrows():
    print(f"  {short(r['Name']):45s} calls {r['Calls']:6d}  avg {r['AverageNs'] / 1e3:7.1f} us  total {r['TotalDurationNs'] / 1e6:8.2f} ms")
tr = pd.read_csv(glob.glob(o + "/**/*kernel_trace.csv", recursive=True)[0]).sort_values("Start_Timestamp")
gap = (tr["Start_Timestamp"].values[1:] - tr["End_Timestamp"].values[:-1])
import numpy as np
g = gap[(gap > -1e6) & (gap < 2e5)]
print(f"gap between consecutive kernels: median {np.median(g) / 1e3:.1f} us, mean {g.mean() / 1e3:.1f} us, p90 {np.percentile(g, 90) / 1e3:.1f} us")
PY
