#!/bin/bash
# usage (GPU box): tools_dev/ab_env.sh "ENV=.." "ENV=.." ...  -- Variant B matrix kernel at BA N = 100k under each environment, in ONE call
for e in "$@"; do
  echo "== $e"
  env $e timeout -k 10 100 python tools_dev/ab_stream.py 100000 B 3 2>&1 | grep "persistent, sorted\|stream<" | cut -c1-260 || exit 1
done
