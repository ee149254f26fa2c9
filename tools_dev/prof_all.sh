#!/bin/bash
# usage (GPU box): tools_dev/prof_all.sh <round tag>   -- the rocprofv3 evidence of one round, summaries under gpurun_out/prof_<tag>_*/
# (copy the summaries into profiles/).  A ticker keeps the call from looking hung while rocprofv3 collects silently.
tag=${1:-rXX}
( while true; do sleep 60; echo "[prof_all] $(date +%T) still collecting"; done ) &
ticker=$!
trap 'kill $ticker 2>/dev/null' EXIT
set -x
PROF_PASSES="trace" PROF_EXTRA_ARGS="" tools_dev/prof.sh ${tag}_bench_default > gpurun_out/prof_${tag}_bench_default.log 2>&1 || exit 1
PROF_STEPS=3 tools_dev/prof.sh ${tag}_target --configs none > gpurun_out/prof_${tag}_target.log 2>&1 || exit 1
PROF_STEPS=3 tools_dev/prof.sh ${tag}_config3 --workload config3 > gpurun_out/prof_${tag}_config3.log 2>&1 || exit 1
PROF_STEPS=20 tools_dev/prof.sh ${tag}_config2 --workload config2 > gpurun_out/prof_${tag}_config2.log 2>&1 || exit 1
# config 5: the Chebyshev kernels of MultiHSD at FULL size (BA N = 50 000) under the counters, ONE impulse block (rows [0, 1024):
# 50 cheby_step dispatches -- all 4 000 of the whole pass outlast a gpurun call under PMC); the kernel-trace of the whole
# run is in ${tag}_bench_default
[ -n "$SKIP_CFG5" ] && exit 0        # (only when its kernels did not change since the last collection -- and then say so in the copied file)
export TMPDIR=/tmp
R=$PWD; out=$R/gpurun_out/prof_${tag}_config5_block; rm -rf $out; mkdir -p $out
for pass in "pmc4 FETCH_SIZE" "pmc5 WRITE_SIZE" "pmc3 GRBM_GUI_ACTIVE"; do
  set -- $pass
  (cd /tmp && rocprofv3 --pmc $2 -d $out/$1 --output-format csv -- python3 $R/tools_dev/cfg5.py 50000 0 1024 > $out/$1.log 2>&1) || { tail -5 $out/$1.log; exit 1; }
done
python3 - $out <<'PY' | tee $out/summary.txt
import csv, glob, collections, sys
out = sys.argv[1]
print("# rocprofv3 PMC passes, command: python3 tools_dev/cfg5.py 50000 0 1024 (MultiHSD, BA N=50000 m=5, 8 scales, K=3, order 50; impulse rows [0, 1024) = one block of 1024 columns)")
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in sorted(glob.glob(out + '/pmc*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': dur[r['Kernel_Name'][:60]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, d in agg.items():
    print(' ', k)
    for c, v in sorted(d.items()): print(f"      {c:20s} {sum(v)/len(v):16.1f} (n={len(v)})")
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
        fs, ws = sum(d['FETCH_SIZE'])/len(d['FETCH_SIZE']), sum(d['WRITE_SIZE'])/len(d['WRITE_SIZE'])
        print(f"      HBM traffic per launch: read {2*fs*1024/1e6:.1f} MB (FETCH_SIZE x 2, gfx950), write {ws*1024/1e6:.1f} MB" + (f", avg {sum(dur[k])/len(dur[k])/1e3:.1f} us" if dur.get(k) else ""))
PY
