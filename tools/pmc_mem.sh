#!/bin/bash
# Memory-side counters of one 1080p frame for library variants (GPU box): tools/pmc_mem.sh "one_frame args" a.so b.so ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ARGS=$1; shift
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so); O=$R/gpurun_out/pmcm/$tag; rm -rf "$O"; mkdir -p "$O"
  export PTMI_LIB=$R/$lib
  i=1
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
    timeout -k 5 200 rocprofv3 --pmc $grp --output-format csv -d "$O/p$i" -o r1 -- python "$R/tools/one_frame.py" $ARGS > "$O/p$i.log" 2>&1 || echo "pass $i failed"
    i=$((i + 1))
  done
  echo "== $lib"
  python - "$O" <<'PY'
import sys, glob, re, pandas as pd
fr = [pd.read_csv(f) for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True)]
pm = pd.concat(fr)
pm["kernel"] = pm["Kernel_Name"].str.replace(r"pt::\(anonymous namespace\)::", "", regex=True).str.replace(r"\(.*", "", regex=True).str.replace("void ", "")
g = pm.groupby(["kernel", "Counter_Name"])["Counter_Value"].sum().unstack()
g = g[g.index.str.startswith("k_")]
out = pd.DataFrame(index=g.index)
out["read_GB"] = 2 * g["FETCH_SIZE"] * 1024 / 1e9
out["write_GB"] = g["WRITE_SIZE"] * 1024 / 1e9
out["L2_hit"] = g["TCC_HIT_sum"] / (g["TCC_HIT_sum"] + g["TCC_MISS_sum"])
out["L2_req_M"] = g["TCC_REQ_sum"] / 1e6
out["wave_Mcyc"] = g["SQ_WAVE_CYCLES"] / 1e6
out["valu_M"] = g["SQ_INSTS_VALU"] / 1e6
out["wait_any"] = g["SQ_WAIT_INST_ANY"] / g["SQ_WAVE_CYCLES"]
pd.set_option("display.width", 200)
print(out.round(3).to_string())
PY
  find "$O" -name "*.csv" -size +5M -delete
done
