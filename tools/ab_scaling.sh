#!/bin/bash
# same-device A/B of library variants on the K-step / tiles-per-CU scaling shapes: tools/ab_scaling.sh ab/a.so ab/b.so
for round in 1 2; do
  for L in "$@"; do
    echo "== $L (round $round)"
    DD_LIB=$PWD/$L timeout 300 python tools/conv_scaling.py 2>&1 | grep -E "^B="
  done
done
