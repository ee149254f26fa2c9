#!/bin/bash
# usage (GPU box): tools_dev/ab_many.sh rounds libA.so libB.so ...  -- Variant B matrix kernel at BA N = 100k, the builds alternately in ONE call
R=$1; shift
for r in $(seq $R); do
  for L in "$@"; do
    echo "== $L"
    HSD_LIB=$PWD/$L timeout -k 10 100 python tools_dev/ab_stream.py 100000 ${AB_VARIANT:-B} 3 2>/dev/null | grep "persistent, sorted\|no stores" | cut -c1-150 || exit 1
  done
done
