#!/bin/bash
# usage: tools_dev/clock.sh "ENV.." ... -- bench args : reports kernel avg ns and effective clock from GRBM_GUI_ACTIVE
export TMPDIR=/tmp
R=$PWD
sets=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do sets+=("$1"); shift; done; shift
i=0
for e in "${sets[@]}"; do
  i=$((i+1)); out=$R/gpurun_out/clk_$i; rm -rf $out; mkdir -p $out
  (cd /tmp && env $e rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $out --output-format csv -- python3 $R/bench.py --cpu-seconds 0 --configs none --steps 10 --warmup 2 "$@" > $out/log.txt 2>&1)
  echo "== $e"
  python3 - $out <<'PY'
import csv, glob, sys, collections
out=sys.argv[1]
dur=collections.defaultdict(list)
for f in glob.glob(out+'/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r['Kernel_Name'][:48]].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
cnt=collections.defaultdict(list)
for f in glob.glob(out+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        cnt[r['Kernel_Name'][:48]].append(float(r['Counter_Value']))
for k in dur:
    if 'pair_kernel' in k:
        d=sum(dur[k])/len(dur[k]); c=sum(cnt[k])/len(cnt[k]) if cnt[k] else 0
        print('  ', k, 'avg_us', round(d/1e3,1), 'clock_GHz', round(c/8/d,3))
PY
done
