"""In-kernel cycle stamps of the persistent all-pairs kernel (a -DHSD_STAMPS=1 -DHSD_DEBUG_HOOKS build of the library, loaded with HSD_LIB;
HSD_DEBUG bit 3): wave 0's cycles per tile in the chunk loop -- of which waiting for a chunk's loads (vmcnt is in order: behind older stores)
and at the chunk barrier --, in the fold + next-tile requests, and in the tile's own stores.  Whole matrix, then rank 0's row block of 8.
usage: HSD_LIB=hsd_amd/libhsd_amd_stamps.so python tools_dev/stamps_run.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsd_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
g = hsd_amd.barabasi_albert(n, 5, 42)
eng = hsd_amd.Engine(0)
eng.set_graph(g.rowptr, g.col)
eng.signature_build_degree(3)
eng.pair_prepare(0, 1, 0, 2)
M = eng.pair_rows_device(); eng.sync()
for dbg in ("0", "8", "10", "9"):
    os.environ["HSD_DEBUG"] = dbg
    print(f"--- whole matrix, HSD_DEBUG={dbg} (8: stamps; +2: no arithmetic; +1: no stores)", file=sys.stderr, flush=True)
    eng.pair_rows_device(0, None, M); eng.sync()
    eng.timer_start(); eng.pair_rows_device(0, None, M); t = eng.timer_stop()
    print(f"    {t:.3f} ms", file=sys.stderr, flush=True)
M.free()
r0, r1 = eng.pair_shard(8, 0)
M = eng.pair_rows_device(r0, r1); eng.sync()
for dbg in ("0", "8", "10", "9"):
    os.environ["HSD_DEBUG"] = dbg
    print(f"--- rank 0 of 8, rows [{r0},{r1}), HSD_DEBUG={dbg}", file=sys.stderr, flush=True)
    eng.pair_rows_device(r0, r1, M); eng.sync()
    eng.timer_start(); eng.pair_rows_device(r0, r1, M); t = eng.timer_stop()
    print(f"    {t:.3f} ms", file=sys.stderr, flush=True)
M.free(); eng.close()
