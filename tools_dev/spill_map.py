"""where a kernel's scratch (spill) instructions sit: python3 tools_dev/spill_map.py file.s <mangled-name-substring>
prints, per basic block that holds scratch operations: loop depth, scratch ops, SADs / fp64 ops / stores in the block."""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
i0 = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + re.escape(sys.argv[2]) + r'\S*:', l))
i1 = next(i for i in range(i0, len(lines)) if 's_endpgm' in lines[i])
body = lines[i0:i1]
blocks, cur = [], ['entry', 0, []]
for l in body:
    if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append(cur)
        d = re.search(r'Depth=(\d+)', l)
        cur = [l.split(':')[0], int(d.group(1)) if d else 0, []]
    else:
        cur[2].append(l.strip())
blocks.append(cur)
tot = collections.Counter()
for name, depth, ins in blocks:
    sc = sum(1 for x in ins if x.startswith('scratch_'))
    sad = sum(1 for x in ins if x.startswith('v_sad') or x.startswith('v_add_f64') or x.startswith('v_fma_f64'))
    st = sum(1 for x in ins if x.startswith('global_store'))
    tot[depth] += sc
    if sc:
        print(f"{name:12s} depth={depth} instrs={len(ins):5d} scratch={sc:4d} arith={sad:4d} stores={st:3d}")
print('scratch ops by loop depth:', dict(tot))
