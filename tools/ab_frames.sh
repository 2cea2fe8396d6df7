#!/bin/bash
# A/B timing of library variants on ONE box, whole frame and rank 0's share of an 8-way split, interleaved:
#   tools/ab_frames.sh build/variants/a.so build/variants/b.so ...
for rep in 1 2; do
  for lib in "$@"; do
    for world in 1 8; do
      PTMI_LIB=$PWD/$lib timeout -k 10 120 python tools/one_frame.py $world 2>/dev/null | grep -v "^B" | sed "s|^|$lib |" || exit 1
    done
  done
done
