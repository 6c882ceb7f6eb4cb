#!/bin/bash
for round in 1 2; do
  for L in "$@"; do
    echo "== $L (round $round)"
    DD_LIB=$PWD/$L timeout 300 python tools/conv_deep.py 2>&1 | grep -E "^(3x3|1x1)"
  done
done
