#!/bin/bash
# A/B of a run-time switch on ONE box, interleaved: tools/ab_env.sh VAR=a VAR=b ...   (whole frame and rank 0's share of 2/4/8-way splits,
# the 82 k mesh at 64 spp, the interactive loop)
for rep in 1 2; do
  for kv in "$@"; do
    export "$kv"
    for world in 1 2 4 8; do
      timeout -k 10 120 python tools/one_frame.py $world 2>/dev/null | grep -v "^B" | sed "s|^|$kv |" || exit 1
    done
    timeout -k 10 120 python tools/one_frame.py 1 0 cornell_mesh:6 64 2>/dev/null | grep -v "^B" | sed "s|^|$kv |" || exit 1
    timeout -k 10 120 python tools/one_frame.py 1 0 cornell_mesh:6 512 2>/dev/null | grep -v "^B" | sed "s|^|$kv |" || exit 1
    timeout -k 10 120 python tools/frame_bench.py 200 2>/dev/null | tail -1 | sed "s|^|$kv |" || exit 1
  done
done
