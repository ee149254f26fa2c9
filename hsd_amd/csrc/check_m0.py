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
    address register within five wait states of a v_readfirstlane / v_readlane that wrote it, and every such store is
    followed by two wait states (its data registers may be rewritten): hazards hipcc pads only for its own instructions.
Exit status 1 with the offending lines otherwise.  Run by __graft_entry__.build() and `make check`."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = os.environ.get("HSD_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
M0_WRITE = re.compile(r"s_mov_b32 m0, (s\d+|vcc_lo|vcc_hi|0x[0-9a-f]+|\d+)$")      # the asm's "s" operand: any scalar register or literal


def disassemble(lib):
    tmp = tempfile.mkdtemp(prefix="hsd_m0_")
    try:
        local = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, local)
        subprocess.check_call([OBJDUMP, "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" in f:
                yield f, subprocess.check_output([OBJDUMP, "-d", os.path.join(tmp, f)], text=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def check(lib):
    bad, kernels, dma = [], 0, 0
    for name, text in disassemble(lib):
        cur, ins = None, []
        funcs = {}
        for line in text.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                cur = m.group(1)
                funcs[cur] = ins = []
                continue
            if cur is None or "//" not in line and not line.startswith("\t"):
                continue
            body = line.split("//")[0].strip()
            if body:
                ins.append(body)
        for fn, ins in funcs.items():
            if "pair_stream_kernel" not in fn:
                continue
            kernels += 1
            # hand-issued vector-memory instructions with a scalar address (LDS-DMA loads, result stores): no scalar register
            # of the address may have been written by the vector ALU (v_readfirstlane / v_readlane) within the five
            # preceding wait states, and a store of more than 64 bits must be followed by two wait states before anything
            # else (its data registers may be rewritten).  hipcc pads only instructions it emitted itself.
            for k, x in enumerate(ins):
                op = x.split()[0]
                if not (op.startswith("global_load_lds") or (op.startswith("global_store_dwordx4") and re.search(r"s\[\d+:\d+\]", x))):
                    continue
                m = re.search(r"s\[(\d+):(\d+)\]", x)
                if not m:
                    continue
                addr = {f"s{r}" for r in range(int(m.group(1)), int(m.group(2)) + 1)}
                waits, j = 0, k - 1
                while j >= 0 and waits < 5:
                    y = ins[j]
                    if y.startswith("s_nop"):
                        waits += int(y.split()[1]) + 1
                    else:
                        if y.startswith(("v_readfirstlane_b32", "v_readlane_b32")) and y.split()[1].rstrip(",") in addr:
                            bad.append(f"{fn}: `{x}` reads {y.split()[1].rstrip(',')} {waits} wait states after `{y}` (needs 5)")
                        waits += 1
                    j -= 1
                if op.startswith("global_store_dwordx4"):
                    nxt = ins[k + 1] if k + 1 < len(ins) else ""
                    if not (nxt.startswith("s_nop") and int(nxt.split()[1]) >= 1):
                        bad.append(f"{fn}: `{x}` is not followed by `s_nop 1` (found `{nxt}`)")
            real = [x for x in ins if not x.startswith("s_nop")]
            for k, x in enumerate(real):
                op = x.split()[0]
                names_m0 = re.search(r"\bm0\b", x) is not None
                if op.startswith("global_load_lds"):
                    dma += 1
                    if k == 0 or not M0_WRITE.match(real[k - 1]):
                        bad.append(f"{fn}: `{x}` is not preceded by `s_mov_b32 m0, sN` (found `{real[k - 1] if k else '-'}`)")
                elif names_m0:
                    if not M0_WRITE.match(x):
                        bad.append(f"{fn}: `{x}` uses m0 outside the LDS-DMA sequence")
                    elif k + 1 >= len(real) or not real[k + 1].startswith("global_load_lds"):
                        bad.append(f"{fn}: `{x}` is not followed by a global_load_lds (found `{real[k + 1] if k + 1 < len(real) else '-'}`)")
    return kernels, dma, bad


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "libhsd_amd.so")
    kernels, dma, bad = check(lib)
    if kernels == 0:
        print("check_m0: no pair_stream_kernel instance found in", lib)
        sys.exit(1)
    if bad:
        print("check_m0: M0 contract violated:")
        for b in bad[:40]:
            print("  ", b)
        sys.exit(1)
    print(f"check_m0: {kernels} pair_stream_kernel instances, {dma} LDS-DMA loads, M0 written only in front of them: ok")
