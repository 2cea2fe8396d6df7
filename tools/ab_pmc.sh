#!/bin/bash
# For each library variant: bench.py timing (twice, interleaved) and one --pmc pass (VALU counters) of a 16-spp frame; traversal kernels only.
# tools/ab_pmc.sh a.so b.so ...
tools/ab_bench.sh "$@"
for lib in "$@"; do
  tag=$(basename $lib .so)
  PMC_PASSES=1 bash tools/pmc_quick.sh $lib $tag > /dev/null 2>&1 || echo "pmc failed for $lib"
  echo "== $lib"; grep "k_closest\|k_any\|waves" gpurun_out/pmcq/$tag/summary.txt | grep -v "1, false"
done
