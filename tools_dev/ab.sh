#!/bin/bash
# usage: tools_dev/ab.sh "ENV1=.. ENV2=.." "ENV.." -- bench args ; prints kernel ms for primary/secondary per env set
sets=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do sets+=("$1"); shift; done; shift
for e in "${sets[@]}"; do
  env $e timeout -k 10 300 python bench.py --cpu-seconds 0 --configs none "$@" 2>&1 | python3 -c "
import sys, json
e = sys.argv[1]
for line in sys.stdin:
    line = line.strip()
    if line.startswith('{'):
        d = json.loads(line)
        s = d.get('secondary') or {}
        sr = s.get('roofline', {})
        print(f\"{e:44s} primary {d['roofline']['kernel_ms']:9.4f} ms hbm {d['roofline']['frac']:.3f} valu {d['roofline']['valu_frac']:.3f} cols/pair {d['roofline'].get('columns_evaluated_per_pair')} | secondary {sr.get('kernel_ms', float('nan')):9.4f} ms hbm {sr.get('frac', float('nan')):.3f} valu {sr.get('valu_frac', float('nan')):.3f} cols/pair {sr.get('columns_evaluated_per_pair')}\")
        break
else:
    print(e, 'FAILED')
" "$e"
done
