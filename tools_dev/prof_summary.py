#!/usr/bin/env python3
"""Summarise a tools_dev/prof.sh output directory (rocprofv3 CSVs) as text: kernel-trace stats + mean PMC values per kernel.
usage: tools_dev/prof_summary.py gpurun_out/prof_<tag> ["command line that was profiled"]"""
import collections
import csv
import glob
import sys

out = sys.argv[1]
args = sys.argv[2] if len(sys.argv) > 2 else ""
KEEP = ("pair_", "khop", "ms_", "cheby", "ring_sum", "ring_gather", "matrix_stats")
print(f"# rocprofv3 summary{', command: ' + args if args else ''}")
print("## kernel-trace --stats (per kernel: calls, average ns, % of GPU time)")
for f in sorted(glob.glob(out + '/trace/**/*kernel_stats.csv', recursive=True)):
    for r in list(csv.DictReader(open(f)))[:18]:
        print(f"  {r.get('Name','')[:100]:100s} calls={r.get('Calls'):>5s} avg_ns={float(r.get('AverageNs')):14.1f} pct={r.get('Percentage')}")
print("## PMC passes (mean per dispatch of each kernel; separate runs per counter group)")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(out + '/pmc*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:90]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE' and r.get('End_Timestamp') and r.get('Start_Timestamp'):
            dur[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, d in agg.items():
    if not any(t in k for t in KEEP):
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
