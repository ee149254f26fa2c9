"""Kernel timeline of one warm hsd_pair_prepare from a rocprofv3 --kernel-trace of tools_dev/prepare_marks.py: start (us from the first kernel of the
call), duration, name; the gaps are host work and synchronisations.   usage: python tools_dev/prepare_timeline.py <dir with *kernel_trace.csv>"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "prep_zp_kernel" in r["Kernel_Name"]]
i0 = idx[2]                                   # third prepare (warm)
i1 = idx[3] if len(idx) > 3 else len(rows)
t0 = int(rows[i0 - 6]["Start_Timestamp"])
for r in rows[i0 - 6:i1 - 6]:
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f}  {r['Kernel_Name'][:90]}")
