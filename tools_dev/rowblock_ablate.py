"""What bounds a rank's ROW BLOCK (rectangles stored once, no mirror) of the headline matrix: the block of rank r of P timed alone on one GPU
with the measurement hooks of the debug library (make -C hsd_amd/csrc debug; HSD_LIB=hsd_amd/libhsd_amd_dbg.so):
  HSD_DEBUG=0   the kernel as shipped
  HSD_DEBUG=2   no arithmetic (every chunk counts zero words): loads, plan and stores only
  HSD_DEBUG=16  every tile stores into the first granules' place: the same store instructions without the HBM write stream
  HSD_DEBUG=18  both
usage: HSD_LIB=hsd_amd/libhsd_amd_dbg.so python tools_dev/rowblock_ablate.py [P] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsd_amd
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
g = hsd_amd.barabasi_albert(n, 5, 42)
eng = hsd_amd.Engine(0)
eng.set_graph(g.rowptr, g.col)
eng.signature_build_degree(3)
eng.pair_prepare(0, 1, 0, 2)
for r in sorted({0, P // 2, P - 1}):
    r0, r1 = eng.pair_shard(P, r)
    M = eng.pair_rows_device(r0, r1); eng.sync()
    for dbg in (("0", "2", "16", "18") if r0 == 0 else ("0", "2")):      # (bit 4 is honoured only for a block that starts at node 0)
        os.environ["HSD_DEBUG"] = dbg
        eng.pair_rows_device(r0, r1, M); eng.sync()
        ts = []
        for _ in range(5):
            eng.timer_start(); eng.pair_rows_device(r0, r1, M); ts.append(eng.timer_stop())
        gb = (r1 - r0) * n * 8 / 1e9
        print(f"P={P} rank {r} rows [{r0},{r1}) {gb:.1f} GB  HSD_DEBUG={dbg:>2s}: best {min(ts):.3f} ms  ({' '.join(f'{t:.2f}' for t in ts)})", flush=True)
    os.environ.pop("HSD_DEBUG", None)
    M.free()
# the whole matrix on one GPU, same hooks, for the comparison
M = eng.pair_rows_device(); eng.sync()
for dbg in ("0", "2", "16", "18"):
    os.environ["HSD_DEBUG"] = dbg
    eng.pair_rows_device(0, None, M); eng.sync()
    ts = []
    for _ in range(4):
        eng.timer_start(); eng.pair_rows_device(0, None, M); ts.append(eng.timer_stop())
    print(f"one GPU, whole matrix  HSD_DEBUG={dbg:>2s}: best {min(ts):.3f} ms  ({' '.join(f'{t:.2f}' for t in ts)})", flush=True)
M.free(); eng.close()
