#!/bin/bash
# Two quick --pmc passes of one 1080p frame for one library variant (GPU box): tools/pmc_quick.sh <lib.so> <tag> [one_frame.py args...]
# -> gpurun_out/pmcq/<tag>/..., summary printed by tools/pmc_quick.py
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
LIB=$1; TAG=$2; shift; shift
ARGS=${@:-1 0 cornell_box 16}
O=$R/gpurun_out/pmcq/$TAG
rm -rf "$O" && mkdir -p "$O"
export PTMI_LIB=$R/$LIB
cd /tmp && export TMPDIR=/tmp
i=1
PASSES=${PMC_PASSES:-2}
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    timeout -k 5 200 rocprofv3 --pmc $grp --output-format csv -d "$O/p$i" -o r1 -- python "$R/tools/one_frame.py" $ARGS > "$O/p$i.log" 2>&1 || { tail -5 "$O/p$i.log"; exit 1; }
    i=$((i + 1)); [ $i -gt $PASSES ] && break
done
python "$R/tools/pmc_quick.py" "$O" | tee "$O/summary.txt"
find "$O" -name "*.csv" -size +5M -delete
