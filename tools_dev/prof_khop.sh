#!/bin/bash
# usage (GPU box): tools_dev/prof_khop.sh <tag> <khop_time.py shapes...>
# rocprofv3 kernel-trace stats of the k-hop sweep A/B (tools_dev/khop_time.py) -> gpurun_out/prof_<tag>/stats.txt
set -o pipefail
tag=$1; shift
R=$PWD
out=$R/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats -d $out/trace --output-format csv -- python3 $R/tools_dev/khop_time.py "$@" > $out/trace.log 2>&1) || { tail -5 $out/trace.log; exit 1; }
f=$(find $out/trace -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY' | tee $out/stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    print(f"{r['Name'][:90]:90s} calls={int(r['Calls']):6d} total_ms={float(r['TotalDurationNs'])/1e6:10.3f} avg_us={float(r['AverageNs'])/1e3:10.1f} pct={float(r['Percentage']):6.2f}")
PY
cat $out/trace.log
