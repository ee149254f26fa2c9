#!/bin/bash
# usage (GPU box): tools_dev/prof.sh <tag> <bench args...>
# rocprofv3 kernel-trace stats of `python3 bench.py <args>` + separate PMC passes (never combined
# with trace domains). Writes gpurun_out/prof_<tag>/summary.txt (+ the raw CSVs).
# PROF_STEPS / PROF_WARMUP (default 5 / 1) set the launches per pass; PROF_PASSES="trace pmc1 pmc2 pmc3 pmc4 pmc5" selects passes.
set -o pipefail
tag=$1; shift
R=$PWD
out=$R/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
BENCH_ARGS="--cpu-seconds 0 --steps ${PROF_STEPS:-5} --warmup ${PROF_WARMUP:-1} $*"
PASSES=${PROF_PASSES:-trace pmc1 pmc2 pmc3 pmc4 pmc5}
run() { name=$1; shift; case " $PASSES " in *" $name "*) ;; *) return 0;; esac
        (cd /tmp && rocprofv3 "$@" -d $out/$name --output-format csv -- python3 $R/bench.py $BENCH_ARGS > $out/$name.log 2>&1) || { tail -5 $out/$name.log; return 1; }
        echo "pass $name done"; }
run trace --kernel-trace --stats || exit 1
run pmc1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY || exit 1
run pmc2 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU || exit 1
run pmc3 --pmc GRBM_GUI_ACTIVE || exit 1
run pmc4 --pmc FETCH_SIZE || exit 1
run pmc5 --pmc WRITE_SIZE || exit 1
python3 - $out "$BENCH_ARGS" <<'PY' | tee $out/summary.txt
import csv, glob, collections, sys
out, args = sys.argv[1], sys.argv[2]
print(f"# rocprofv3 summary, command: python3 bench.py {args}")
print("## kernel-trace --stats (per kernel: calls, average ns, % of GPU time)")
for f in sorted(glob.glob(out + '/trace/**/*kernel_stats.csv', recursive=True)):
    for r in list(csv.DictReader(open(f)))[:16]:
        print(f"  {r.get('Name','')[:96]:96s} calls={r.get('Calls'):>4s} avg_ns={float(r.get('AverageNs')):14.1f} pct={r.get('Percentage')}")
print("## PMC passes (mean per dispatch of each kernel; separate runs per counter group)")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(out + '/pmc*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'][:80]][r['Counter_Name']].append(float(r['Counter_Value']))
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE' and r.get('End_Timestamp') and r.get('Start_Timestamp'):
            dur[r['Kernel_Name'][:80]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, d in agg.items():
    if 'pair_kernel' not in k and 'khop' not in k and 'cheby' not in k:
        continue
    print(' ', k)
    for c, v in sorted(d.items()):
        print(f"      {c:24s} {sum(v)/len(v):18.1f}   (n={len(v)})")
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
        fs, ws = sum(d['FETCH_SIZE'])/len(d['FETCH_SIZE']), sum(d['WRITE_SIZE'])/len(d['WRITE_SIZE'])
        print(f"      HBM traffic per launch: read {2*fs*1024/1e6:.1f} MB (FETCH_SIZE KiB x 2: gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md HBM), write {ws*1024/1e6:.1f} MB")
    if dur.get(k) and 'GRBM_GUI_ACTIVE' in d:
        ns = sum(dur[k]) / len(dur[k]); gui = sum(d['GRBM_GUI_ACTIVE']) / len(d['GRBM_GUI_ACTIVE'])
        print(f"      effective clock: GRBM_GUI_ACTIVE / 8 XCDs / {ns/1e3:.1f} us = {gui/8/ns:.3f} GHz")
PY
