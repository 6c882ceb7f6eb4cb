#!/bin/bash
# same-device A/B of library variants: tools/ab_conv.sh ab/v4.so ab/v7.so ...
for round in 1 2; do
  for L in "$@"; do
    echo "== $L (round $round)"
    DD_LIB=$PWD/$L timeout 300 python tools/bench_conv.py 2>&1 | grep -E "^3x3 (320->320 |640|1280->1280 @16|512)|^1x1 320->2560 geglu |^1x1 320->320 \+bias\+res " | grep -v small
  done
done
