"""Kernel timeline (start, duration, name; gaps = host work / synchronisation / copies) of ONE warm hsd_pair_topk call at the target, out of a rocprofv3
kernel trace of this script.   run:   rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools_dev/topk_timeline.py
       print: python3 tools_dev/topk_timeline.py --print <dir>"""
import csv, glob, os, sys, time
if len(sys.argv) > 2 and sys.argv[1] == "--print":
    rows = []
    for f in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last call stands alone: the script sleeps 60 ms in front of it
    i0 = max(i for i in range(1, len(rows)) if int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 40_000_000)
    t0 = int(rows[i0]["Start_Timestamp"])
    busy = 0
    for r in rows[i0:]:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); busy += d
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us +{d / 1e3:8.1f}  {r['Kernel_Name'][:100]}")
    print(f"span {(int(rows[-1]['End_Timestamp']) - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsd_amd
n = 100000
g = hsd_amd.barabasi_albert(n, 5, 42)
eng = hsd_amd.Engine(0)
eng.set_graph(g.rowptr, g.col)
eng.signature_build_degree(3)
eng.pair_prepare(0, 1, 0, 2)
for rep in range(4):
    if rep == 3:
        time.sleep(0.06)
    eng.sync(); t0 = time.perf_counter()
    idx, val = eng.pair_topk(20)
    eng.sync()
    print(f"topk call {rep}: {(time.perf_counter() - t0) * 1e3:.3f} ms", file=sys.stderr, flush=True)
eng.close()
