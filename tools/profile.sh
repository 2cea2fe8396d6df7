#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_*: one kernel-trace pass of a bench.py step and separate --pmc passes (never combined with other
# trace domains; FETCH_SIZE and WRITE_SIZE each need a pass of their own on gfx950).  Run on the GPU box:
#   bash tools/profile.sh <tag> [bench.py arguments, e.g. --config mesh82k --spp 8]      (default tag r04_cornell, --spp 43 for the PMC passes)
# (every pass with --pipelines 1: batches one after another, so that kernel durations and counters are not those of two overlapping pipelines)
# then  python tools/profile_summary.py gpurun_out/prof_<tag> <tag>
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04_cornell}; shift || true
ARGS="$@"
case "$ARGS" in *--spp*) PMC_ARGS="$ARGS";; *) PMC_ARGS="$ARGS --spp 43";; esac
O=$R/gpurun_out/prof_$TAG
rm -rf "$O" && mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
echo "$ARGS" > "$O/args.txt"
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -o r1 -- python "$R/bench.py" $ARGS --pipelines 1 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-ms > "$O/kt.log" 2>&1
echo "kernel trace done"
i=1
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
           "FETCH_SIZE" \
           "WRITE_SIZE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    timeout -k 5 400 rocprofv3 --pmc $grp --output-format csv -d "$O/pmc$i" -o r1 -- python "$R/bench.py" $PMC_ARGS --pipelines 1 --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-ms > "$O/pmc$i.log" 2>&1 || echo "pmc pass $i ($grp) failed"
    echo "pmc pass $i done"
    i=$((i + 1))
done
find "$O" -name "*.csv" -size +20M -delete   # per-dispatch traces are not needed, the stats and counter tables are
ls "$O"
