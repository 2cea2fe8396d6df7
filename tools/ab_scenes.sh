#!/bin/bash
# A/B timing of library variants on the mesh scenes (BVH in global memory), one box, interleaved: tools/ab_scenes.sh [stack_lds] -- a.so b.so ...
SL=0
if [ "$2" == "--" ]; then SL=$1; shift; shift; fi
for rep in 1 2; do
  for lib in "$@"; do
    for sc in cornell_mesh:6 cornell_mesh:7 cornell_spheres:4; do
      PTMI_LIB=$PWD/$lib timeout -k 10 120 python tools/one_frame.py 1 0 $sc 8 1 $SL 2>/dev/null | grep -v "^B" | sed "s|^|$lib |" || exit 1
    done
  done
done
