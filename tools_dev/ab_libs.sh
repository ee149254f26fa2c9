#!/bin/bash
# usage (GPU box): tools_dev/ab_libs.sh libA.so libB.so [rounds=2]  -- Variant B matrix kernel at BA N = 100k, the two builds alternately in ONE call
A=$1; B=$2; R=${3:-2}
for r in $(seq $R); do
  for L in $A $B; do
    echo "== $L"
    HSD_LIB=$PWD/$L timeout -k 10 100 python tools_dev/ab_stream.py 100000 B 3 2>/dev/null | grep "persistent, sorted\|no stores" | cut -c1-150 || exit 1
  done
done
