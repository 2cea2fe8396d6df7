#!/bin/bash
# per-launch durations of the LAST frame's late bounces for library variants (1/8 share), under rocprofv3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in "$@"; do
  n=$(basename $lib .so)
  PTMI_LIB=$R/$lib timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/late_$n -o r1 -- python $R/tools/one_frame.py 8 > $R/gpurun_out/late_$n.log 2>&1 || exit 1
  python $R/tools/kernel_times.py $R/gpurun_out/late_$n/r1_kernel_trace.csv --timeline > $R/gpurun_out/late_$n.txt
  echo "== $n"; grep -v "^B\|^W\|^E" $R/gpurun_out/late_$n.log | tail -1; tail -30 $R/gpurun_out/late_$n.txt | awk '{printf "%s %s | ", $4, $7} END {print ""}'
done
