#!/usr/bin/env python3
"""Summarise a tools_dev/prof.sh output directory (rocprofv3 CSVs) as text.

  ## kernel-trace --stats   rocprofv3's own per-kernel table (calls, average)
  ## kernel trace by launch shape   the same dispatches grouped by (kernel, grid size): calls, first (cold) call, min / median /
                            mean / max, and the mean without the first call -- an all-pairs kernel serves the matrix launches AND
                            the k-nearest sweeps / strips of the same instantiation; they differ in grid size, and the figure a
                            roofline fraction is quoted on is the matrix launches' (the largest grid of that kernel)
  ## PMC passes             mean per dispatch, grouped by (kernel, grid size) in the same way (separate runs per counter group);
                            HBM bytes = FETCH_SIZE KiB x 2 (gfx950 counts a 128-B request at 64 B, MI355X_MICROARCH.md) + WRITE_SIZE KiB

usage: tools_dev/prof_summary.py gpurun_out/prof_<tag> ["command line that was profiled"]"""
import collections
import csv
import glob
import statistics
import sys

KEEP = ("pair_", "khop", "ms_", "cheby", "ring_sum", "ring_gather", "matrix_stats", "fillBuffer")


def trace_groups(out):
    """{(kernel, grid): [durations ns in dispatch order]} of the kernel-trace pass."""
    rows = []
    for f in sorted(glob.glob(out + '/trace/**/*kernel_trace.csv', recursive=True)):
        for r in csv.DictReader(open(f)):
            grid = int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y') or 1) * int(r.get('Grid_Size_Z') or 1)
            rows.append((int(r['Start_Timestamp']), r['Kernel_Name'], grid, int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
    rows.sort()
    g = collections.OrderedDict()
    for _, k, grid, d in rows:
        g.setdefault((k, grid), []).append(d)
    return g


def pmc_groups(out):
    """{(kernel, grid): {counter: [values]}}, {(kernel, grid): [durations ns]} of the PMC passes."""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in sorted(glob.glob(out + '/pmc*/**/*counter_collection.csv', recursive=True)):
        for r in csv.DictReader(open(f)):
            key = (r['Kernel_Name'], int(r['Grid_Size']))
            agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
            if r['Counter_Name'] == 'GRBM_GUI_ACTIVE' and r.get('End_Timestamp') and r.get('Start_Timestamp'):
                dur[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    return agg, dur


def main():
    out = sys.argv[1]
    args = sys.argv[2] if len(sys.argv) > 2 else ""
    print(f"# rocprofv3 summary{', command: ' + args if args else ''}")
    print("## kernel-trace --stats (per kernel: calls, average ns, % of GPU time)")
    for f in sorted(glob.glob(out + '/trace/**/*kernel_stats.csv', recursive=True)):
        for r in list(csv.DictReader(open(f)))[:18]:
            print(f"  {r.get('Name','')[:100]:100s} calls={r.get('Calls'):>5s} avg_ns={float(r.get('AverageNs')):14.1f} pct={r.get('Percentage')}")
    print("## kernel trace by launch shape (kernel, grid size in work-items; ns): calls, first = the cold first dispatch of that shape, then min / median / mean / max"
          " over all of them and the mean WITHOUT the first")
    for (k, grid), d in trace_groups(out).items():
        if not any(t in k for t in KEEP) or (len(d) < 2 and max(d) < 50000) or ("fillBuffer" in k and max(d) < 1000000):
            continue
        warm = d[1:] if len(d) > 1 else d
        print(f"  {k[:84]:84s} grid={grid:>9d} calls={len(d):>4d} first={d[0]:>11d} min={min(d):>11d} median={int(statistics.median(d)):>11d} "
              f"mean={sum(d) / len(d):>13.1f} max={max(d):>11d} mean_warm={sum(warm) / len(warm):>13.1f}")
    print("## PMC passes (mean per dispatch, by kernel and grid size; separate runs per counter group)")
    agg, dur = pmc_groups(out)
    for (k, grid), d in agg.items():
        if not any(t in k for t in KEEP) or "fillBuffer" in k:
            continue
        print(f"  {k[:90]}  grid={grid}")
        for c, v in sorted(d.items()):
            print(f"      {c:24s} {sum(v)/len(v):18.1f}   (n={len(v)})")
        if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
            fs, ws = sum(d['FETCH_SIZE'])/len(d['FETCH_SIZE']), sum(d['WRITE_SIZE'])/len(d['WRITE_SIZE'])
            print(f"      HBM traffic per launch: read {2*fs*1024/1e6:.1f} MB (FETCH_SIZE KiB x 2: gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md HBM), write {ws*1024/1e6:.1f} MB")
        if dur.get((k, grid)) and 'GRBM_GUI_ACTIVE' in d and sum(dur[(k, grid)]) / len(dur[(k, grid)]) > 100000:
            ns = sum(dur[(k, grid)]) / len(dur[(k, grid)]); gui = sum(d['GRBM_GUI_ACTIVE']) / len(d['GRBM_GUI_ACTIVE'])
            print(f"      effective clock: GRBM_GUI_ACTIVE / 8 XCDs / {ns/1e3:.1f} us = {gui/8/ns:.3f} GHz")


if __name__ == "__main__":
    main()
