#!/bin/bash
# A/B timing of library variants on ONE box: tools/ab_bench.sh build/variants/a.so build/variants/b.so ...   (bench.py --steps 4, twice each, interleaved)
for rep in 1 2; do
  for lib in "$@"; do
    PTMI_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$lib', round(d['ms_per_step'],2), 'ms/step', round(d['roofline']['avg_launch_ms'],3))"
  done
done
