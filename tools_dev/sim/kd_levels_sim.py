"""dev (CPU, numpy): the level-by-level form of the k-d node order that csrc/tkorder.hip runs on the device -- per level: extents of every
range, one key per position (range index, value of the range's widest feature; leaves keep their order), ONE stable sort over all positions --
against the recursive form (median split by selection, range after range) that the host threads run (HSD_TOPK_ORDER=0).  With distinct feature
values both give the same leaves (as sets of nodes); the level lists depend on n alone.  usage: python3 tools_dev/sim/kd_levels_sim.py [n ...]"""
import sys
import numpy as np

LEAF = 128


def levels(n):
    """Boundary lists of every level that still has a range longer than a leaf, and the final leaves' boundaries."""
    b, out = [0, n], []
    while True:
        split, nb = False, [0]
        for lo, hi in zip(b[:-1], b[1:]):
            if hi - lo > LEAF:
                split = True
                mid = lo + ((hi - lo) // 2 + LEAF - 1) // LEAF * LEAF
                if mid >= hi:
                    mid = lo + (hi - lo) // 2
                nb.append(mid)
            nb.append(hi)
        if not split:
            return out, b
        out.append(b)
        b = nb


def order_by_levels(F):
    nf, n = F.shape
    order = np.arange(n)
    lv, leaves = levels(n)
    for b in lv:
        b = np.asarray(b)
        rng_of = np.searchsorted(b, np.arange(n), side="right") - 1
        low = np.zeros(n)
        for lo, hi in zip(b[:-1], b[1:]):
            if hi - lo <= LEAF:
                low[lo:hi] = np.arange(hi - lo)
            else:
                sub = F[:, order[lo:hi]]
                low[lo:hi] = sub[int(np.argmax(sub.max(1) - sub.min(1)))]
        order = order[np.lexsort((low, rng_of))]              # (lexsort is stable: ties keep the order they had)
    return order, leaves


def order_by_recursion(F):
    nf, n = F.shape
    idx = np.arange(n)
    todo = [(0, n)]
    while todo:
        lo, hi = todo.pop()
        if hi - lo <= LEAF:
            continue
        sub = F[:, idx[lo:hi]]
        best = int(np.argmax(sub.max(1) - sub.min(1)))
        mid = lo + ((hi - lo) // 2 + LEAF - 1) // LEAF * LEAF
        if mid >= hi:
            mid = lo + (hi - lo) // 2
        idx[lo:hi] = idx[lo:hi][np.lexsort((idx[lo:hi], sub[best]))]
        todo += [(lo, mid), (mid, hi)]
    return idx


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for n in [int(a) for a in sys.argv[1:]] or [100, 128, 129, 700, 2222, 24077, 100000]:
        F = rng.random((4, n))
        d, leaves = order_by_levels(F)
        h = order_by_recursion(F)
        assert sorted(d) == list(range(n))
        same = all(set(d[a:b]) == set(h[a:b]) for a, b in zip(leaves[:-1], leaves[1:]))
        print(f"n={n}: {len(levels(n)[0])} levels, {len(leaves) - 1} leaves (longest {max(np.diff(leaves))}), same leaves as the recursion: {same}")
        assert same
