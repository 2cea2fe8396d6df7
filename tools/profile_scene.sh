#!/bin/bash
# rocprofv3 passes for one scene class:  bash tools/profile_scene.sh <tag> <scene[:level]> <spp>
# kernel-trace stats, then separate --pmc passes (FETCH_SIZE and WRITE_SIZE each alone, L2 hit/miss, SQ issue counters) over
# tools/one_frame.py, then the SURVEY 8(d) roofline against the oracle's node/triangle counters (tools/scene_roofline.py).
# Output: gpurun_out/prof_<tag>/ ; summarise with  python tools/scene_summary.py gpurun_out/prof_<tag> <tag>
set -e
TAG=$1; SCENE=$2; SPP=$3
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_$TAG
rm -rf "$O" && mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -o r1 -- python "$R/tools/one_frame.py" 1 0 "$SCENE" "$SPP" > "$O/kt.log" 2>&1
echo "kernel trace done"
i=1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
    timeout -k 5 300 rocprofv3 --pmc $grp --output-format csv -d "$O/pmc$i" -o r1 -- python "$R/tools/one_frame.py" 1 0 "$SCENE" "$SPP" > "$O/pmc$i.log" 2>&1 || echo "pmc pass $i ($grp) failed"
    echo "pmc pass $i done"
    i=$((i + 1))
done
cd "$R"
timeout -k 5 300 python tools/scene_roofline.py "$SCENE" "$SPP" > "$O/roofline.json" 2> "$O/roofline.err" || echo "roofline failed"
find "$O" -name "*.csv" -size +20M -delete
tail -1 "$O/roofline.json"
