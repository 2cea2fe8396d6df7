#!/bin/bash
# A/B of library variants over a list of one_frame.py configurations, interleaved: tools/ab_any.sh "cfg1;cfg2;..." a.so b.so ...
IFS=';' read -ra CFGS <<< "$1"; shift
for rep in 1 2; do
  for lib in "$@"; do
    for cfg in "${CFGS[@]}"; do
      PTMI_LIB=$PWD/$lib timeout -k 10 120 python tools/one_frame.py $cfg 2>/dev/null | grep -v "^B" | sed "s|^|$lib |" || exit 1
    done
  done
done
