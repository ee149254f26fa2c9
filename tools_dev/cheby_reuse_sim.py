#!/usr/bin/env python3
"""How much of the Chebyshev step's neighbour-row gather could an XCD's L2 serve under different ROW ORDERS?
LRU simulation of one gather stream (every row gathers itself and its neighbours) on BASELINE's config-5 graph
(BA N = 50 000, m = 5).  Result (round 2): 9-14 % at 2-KiB rows (256 impulses per block), 25-29 % at 512-B rows,
whatever the order (as generated / reverse Cuthill-McKee / BFS from the largest hub / degree-sorted / grouped by hub):
a scale-free graph has no row order that makes neighbour rows recur within 4 MiB, so the step's traffic
(measured 5.4 x the once-per-row bytes, profiles/r02_config5_cheby_n20000_pmc.txt) is inherent to the SpMM, and the
kernel is bound by the Infinity-Cache gather rate it already reaches (~7.5 TB/s)."""
import collections
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hsd_amd.graph import barabasi_albert          # noqa: E402
from scipy.sparse import csr_matrix                # noqa: E402
from scipy.sparse.csgraph import reverse_cuthill_mckee   # noqa: E402

g = barabasi_albert(int(sys.argv[1]) if len(sys.argv) > 1 else 50000, 5, 42)
n, rp, col = g.n_node, g.rowptr, g.col


def hit_rate(order, cap):
    lru, hits, tot = collections.OrderedDict(), 0, 0
    for i in order:
        for j in list(col[rp[i]:rp[i + 1]]) + [i]:
            tot += 1
            if j in lru:
                hits += 1
                lru.move_to_end(j)
            else:
                lru[j] = 1
                if len(lru) > cap:
                    lru.popitem(last=False)
    return hits / tot


A = csr_matrix((np.ones(len(col)), col, rp), shape=(n, n))
seen = np.zeros(n, bool)
bfs, dq = [], collections.deque([int(np.argmax(g.degree))])
seen[dq[0]] = True
while dq:
    u = dq.popleft()
    bfs.append(u)
    for v in col[rp[u]:rp[u + 1]]:
        if not seen[v]:
            seen[v] = True
            dq.append(v)
big = np.array([col[rp[i]:rp[i + 1]][np.argmax(g.degree[col[rp[i]:rp[i + 1]]])] for i in range(n)])
orders = {"as generated (age)": np.arange(n), "reverse Cuthill-McKee": reverse_cuthill_mckee(A, symmetric_mode=True),
          "degree descending": np.argsort(-g.degree, kind="stable"), "BFS from the largest hub": np.array(bfs + [i for i in range(n) if not seen[i]]),
          "grouped by highest-degree neighbour": np.lexsort((np.arange(n), big))}
for cap, label in ((2000, "4 MiB of 2-KiB rows (block of 256 impulses)"), (8000, "4 MiB of 512-B rows (block of 64 impulses)")):
    print(label)
    for k, o in orders.items():
        print(f"   {k:40s} L2 hit rate {hit_rate(o, cap):.3f}")
