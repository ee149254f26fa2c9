#!/usr/bin/env python3
"""Build-time contract check of the hand-written LDS-DMA sequence in pair_stream_kernel (csrc/pair.hip, glds16_s).

glds16_s writes M0 from inline asm (`s_mov_b32 m0, <sgpr>; s_nop 0; global_load_lds_dwordx4 ...`) and hipcc does not accept
M0 as a clobber, so nothing in the compiler's view says that M0 changes there.  That is safe only while the compiler keeps
nothing of its own in M0 inside these kernels.  This script disassembles the gfx950 code objects of libhsd_amd.so and, for
every pair_stream_kernel instance, asserts:
  * M0 is written only by `s_mov_b32 m0, <scalar>` and each such write is followed (s_nop aside) by a global_load_lds_dwordx4;
  * every global_load_lds_dwordx4 is preceded (s_nop aside) by such a write;
  * no other instruction names m0 (s_movrel*, s_sendmsg, v_interp, DS with M0 ... would read a value we clobbered);
  * the hand-issued result stores (global_store_dwordx4 with a scalar base) and the LDS-DMA loads never read a scalar
    address register within five wait states of ANY vector-ALU instruction that wrote it (lane reads, compares into an SGPR
    pair, carry-outs), on any control-flow path (the walk follows branches backwards), and every such store is followed by
    two wait states (its data registers may be rewritten): hazards hipcc pads only for its own instructions.
Exit status 1 with the offending lines otherwise.  Run by __graft_entry__.build() and `make check`."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

M0_WRITE = re.compile(r"s_mov_b32 m0, (s\d+|vcc_lo|vcc_hi|0x[0-9a-f]+|\d+)$")      # the asm's "s" operand: any scalar register or literal
HAZARD_WAITS = 5          # VALU write of an SGPR -> vector-memory read of it as an address
# vector-ALU instructions whose scalar destination is their SECOND operand (carry / scale flag); compares and lane reads write the first
SDST_SECOND = ("v_add_co_", "v_sub_co_", "v_subrev_co_", "v_addc_co_", "v_subb_co_", "v_subbrev_co_", "v_div_scale_", "v_mad_u64_u32", "v_mad_i64_i32")


def find_objdump():
    """llvm-objdump: $HSD_OBJDUMP, the ROCm LLVM directory, what the ROCm clang names as its own, or PATH."""
    cands = [os.environ.get("HSD_OBJDUMP"), os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin", "llvm-objdump")]
    for clang in (os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin", "clang"), shutil.which("amdclang") or ""):
        if clang and os.path.exists(clang):
            try:
                cands.append(subprocess.check_output([clang, "--print-prog-name=llvm-objdump"], text=True, stderr=subprocess.DEVNULL).strip())
            except (OSError, subprocess.CalledProcessError):
                pass
    cands += [shutil.which("llvm-objdump"), shutil.which("llvm-objdump-18"), shutil.which("llvm-objdump-17")]
    for c in cands:
        if c and os.path.isfile(c) and os.access(c, os.X_OK):
            return c
    return None


def disassemble(objdump, lib):
    tmp = tempfile.mkdtemp(prefix="hsd_m0_")
    try:
        local = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, local)
        subprocess.check_call([objdump, "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" in f:
                yield f, subprocess.check_output([objdump, "-d", os.path.join(tmp, f)], text=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def sregs(tok):
    """scalar registers named by one operand token: s5 -> {s5}, s[4:7] -> {s4..s7}, vcc -> {vcc_lo, vcc_hi}"""
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return {f"s{r}" for r in range(int(m.group(1)), int(m.group(2)) + 1)}
    if re.fullmatch(r"s\d+", tok):
        return {tok}
    if tok == "vcc":
        return {"vcc_lo", "vcc_hi"}
    if tok in ("vcc_lo", "vcc_hi"):
        return {tok}
    return set()


def valu_sgpr_writes(x):
    """scalar registers a vector-ALU instruction writes (any of them, not only the lane reads: compares into an SGPR pair, carry-outs ...)"""
    parts = x.split(None, 1)
    op = parts[0]
    if not op.startswith("v_") or len(parts) < 2:
        return set()
    ops = [o.strip() for o in parts[1].split(",")]
    w = set()
    if op.startswith(("v_readfirstlane_b32", "v_readlane_b32")) or (op.startswith("v_cmp") and not op.startswith("v_cmpx")):
        w |= sregs(ops[0])
        if op.startswith("v_cmp") and op.endswith("_e32") and not sregs(ops[0]):
            w |= {"vcc_lo", "vcc_hi"}                              # (some printers omit the implicit vcc)
    elif op.startswith(SDST_SECOND) and len(ops) > 1:
        w |= sregs(ops[1])
    return w


class Func:
    """instructions of one kernel with their addresses and the predecessor relation of the flat list (fall-through + branches)"""

    def __init__(self, name, base):
        self.name, self.base, self.ins, self.addr = name, base, [], []
        self.preds = None

    def build(self):
        index = {a: k for k, a in enumerate(self.addr)}
        n = len(self.ins)
        self.preds = [[] for _ in range(n)]
        self.unknown_target = []
        for k, x in enumerate(self.ins):
            op = x.split()[0]
            falls = not (op in ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64") or op.startswith("s_endpgm"))
            if falls and k + 1 < n:
                self.preds[k + 1].append(k)
            if op in ("s_branch",) or op.startswith("s_cbranch"):
                t = self.targets.get(k)
                if t is None or t not in index:
                    self.unknown_target.append(x)
                else:
                    self.preds[index[t]].append(k)
            elif op in ("s_setpc_b64", "s_swappc_b64"):
                self.unknown_target.append(x)                   # indirect: every instruction could be a target
        self.is_target = [len(p) > 1 or (len(p) == 1 and p[0] != k - 1) for k, p in enumerate(self.preds)]


def parse(text):
    funcs, cur = [], None
    for line in text.split("\n"):
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
        if m:
            cur = Func(m.group(2), int(m.group(1), 16))
            cur.targets = {}
            funcs.append(cur)
            continue
        if cur is None or "//" not in line:
            continue
        body, _, comment = line.partition("//")
        body = body.strip()
        ma = re.match(r"\s*([0-9A-Fa-f]+):", comment)
        if not body or not ma:
            continue
        cur.ins.append(body)
        cur.addr.append(int(ma.group(1), 16))
        mt = re.search(r"<(.+)\+0x([0-9a-f]+)>\s*$", comment) or re.search(r"<(.+)>\s*$", comment)
        if mt and (body.startswith("s_branch") or body.startswith("s_cbranch")):
            off = int(mt.group(2), 16) if mt.lastindex and mt.lastindex >= 2 else 0
            if mt.group(1) == cur.name:
                cur.targets[len(cur.ins) - 1] = cur.base + off
    return funcs


def hazard_walk(fn, k, addr, x, bad):
    """every path of up to HAZARD_WAITS wait states that ends in front of instruction k: no vector-ALU write of a register in addr"""
    seen = {}
    stack = [(p, 0) for p in fn.preds[k]]
    while stack:
        j, waits = stack.pop()
        if waits >= HAZARD_WAITS or seen.get(j, HAZARD_WAITS + 1) <= waits:
            continue
        seen[j] = waits
        y = fn.ins[j]
        if y.startswith("s_nop"):
            w = int(y.split()[1]) + 1
        else:
            hit = valu_sgpr_writes(y) & addr
            if hit:
                bad.append(f"{fn.name}: `{x}` reads {sorted(hit)[0]} {waits} wait states after `{y}` (needs {HAZARD_WAITS})")
            w = 1
        for p in fn.preds[j]:
            stack.append((p, waits + w))


def check(lib, objdump):
    bad, kernels, dma = [], 0, 0
    for name, text in disassemble(objdump, lib):
        for fn in parse(text):
            if "pair_stream_kernel" not in fn.name:
                continue
            kernels += 1
            fn.build()
            ins = fn.ins
            if fn.unknown_target:
                bad.append(f"{fn.name}: branch with a target this script cannot resolve (`{fn.unknown_target[0]}`): the hazard walk would be unsound")
                continue
            # hand-issued vector-memory instructions with a scalar address (LDS-DMA loads, result stores): no scalar register
            # of the address may have been written by the vector ALU within the five preceding wait states ON ANY PATH (the walk
            # follows the control-flow graph backwards: fall-through and every branch that targets an instruction on the way),
            # and a store of more than 64 bits must be followed by two wait states before anything else (its data registers may be
            # rewritten) -- and that s_nop must not be a branch target.  hipcc pads only instructions it emitted itself.
            for k, x in enumerate(ins):
                op = x.split()[0]
                if not (op.startswith("global_load_lds") or (op.startswith("global_store_dwordx4") and re.search(r"\bs\[\d+:\d+\]|\bvcc\b", x))):
                    continue
                addr = set()
                for tok in re.findall(r"s\[\d+:\d+\]|\bvcc\b", x):
                    addr |= sregs(tok)
                if addr:
                    hazard_walk(fn, k, addr, x, bad)
                if op.startswith("global_store_dwordx4"):
                    nxt = ins[k + 1] if k + 1 < len(ins) else ""
                    if not (nxt.startswith("s_nop") and int(nxt.split()[1]) >= 1):
                        bad.append(f"{fn.name}: `{x}` is not followed by `s_nop 1` (found `{nxt}`)")
                    elif fn.is_target[k + 1]:
                        bad.append(f"{fn.name}: the `s_nop` behind `{x}` is a branch target")
            # M0: written only by the s_mov_b32 of glds16_s, read only by the global_load_lds behind it; nothing may enter the
            # sequence from elsewhere (no branch target between the write and the load)
            real = [k for k, x in enumerate(ins) if not x.startswith("s_nop")]
            for q, k in enumerate(real):
                x = ins[k]
                op = x.split()[0]
                names_m0 = re.search(r"\bm0\b", x) is not None
                prev = ins[real[q - 1]] if q else "-"
                if op.startswith("global_load_lds"):
                    dma += 1
                    if q == 0 or not M0_WRITE.match(prev):
                        bad.append(f"{fn.name}: `{x}` is not preceded by `s_mov_b32 m0, sN` (found `{prev}`)")
                    elif any(fn.is_target[t] for t in range(real[q - 1] + 1, k + 1)):
                        bad.append(f"{fn.name}: a branch enters between `{prev}` and `{x}`")
                elif names_m0:
                    nxt = ins[real[q + 1]] if q + 1 < len(real) else "-"
                    if not M0_WRITE.match(x):
                        bad.append(f"{fn.name}: `{x}` uses m0 outside the LDS-DMA sequence")
                    elif not nxt.startswith("global_load_lds"):
                        bad.append(f"{fn.name}: `{x}` is not followed by a global_load_lds (found `{nxt}`)")
    return kernels, dma, bad


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "libhsd_amd.so")
    objdump = find_objdump()
    if objdump is None:
        # The check is the only guard of the hand-padded hazards: a build whose kernels changed must not pass without it.
        print("check_m0: llvm-objdump not found (looked at $HSD_OBJDUMP, $ROCM_PATH/lib/llvm/bin, clang --print-prog-name, PATH); "
              "set HSD_OBJDUMP=/path/to/llvm-objdump, or HSD_SKIP_ASM_CHECK=1 to build without the check of the hand-issued instructions")
        sys.exit(0 if os.environ.get("HSD_SKIP_ASM_CHECK") == "1" else 1)
    kernels, dma, bad = check(lib, objdump)
    if kernels == 0:
        print("check_m0: no pair_stream_kernel instance found in", lib)
        sys.exit(1)
    if bad:
        print("check_m0: M0 contract violated:")
        for b in bad[:40]:
            print("  ", b)
        sys.exit(1)
    print(f"check_m0: {kernels} pair_stream_kernel instances, {dma} LDS-DMA loads, M0 written only in front of them, "
          f"hazards checked along every control-flow path: ok")
