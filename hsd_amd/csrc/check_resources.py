#!/usr/bin/env python3
"""Build-time guard on the register / scratch / LDS contract the all-pairs kernels are designed around.

The persistent kernels run TWO workgroups of four waves per CU, i.e. two waves per SIMD: that holds only while a wave needs at most
256 vector registers (arch + accumulation) and the workgroup at most half of the CU's 160 KiB of LDS; and the headline kernel's time
assumes its inner loop touches no scratch -- it sits at 255 of 256 registers with zero spills, and three documented experiments fell off
that cliff (464 / 800 / 1 300 spilled registers, 14.7 -> 22+ ms with every test green).  A compiler point release can do the same silently.
This script reads the kernel descriptors' metadata (llvm-readelf --notes: vgpr_count, agpr_count, vgpr_spill_count, sgpr_spill_count,
private_segment_fixed_size, group_segment_fixed_size, max_flat_workgroup_size) of every gfx950 code object in libhsd_amd.so and fails
the build when a kernel leaves its contract (CONTRACT below: first matching rule wins).

    python3 check_resources.py [libhsd_amd.so]          exit status 1 with the offending kernels; --list prints every kernel's figures

Run by __graft_entry__.build() and `make check`; tests/test_asm_check.py exercises the rules on synthetic metadata."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LDS_PER_CU = 160 * 1024
VGPR_FILE_PER_SIMD = 512          # gfx950: unified arch + accumulation registers per lane and SIMD

# (regex on the demangled-ish name, rule).  Rule keys:
#   max_regs      vgpr_count + agpr_count at most (256 <=> two waves per SIMD)
#   spills        False: vgpr_spill_count == sgpr-to-memory spills == 0 and no scratch (private_segment_fixed_size == 0)
#   max_scratch   bytes of scratch per lane at most (kernels that are allowed a little)
#   max_lds       static LDS bytes at most (the persistent kernels' LDS is dynamic, sized by the host: its budget -- two workgroups
#                 per CU -- is a static_assert on stream_lds_bytes() in csrc/pair.hip, i.e. checked by the compiler itself)
# KIND: 0 raw fp64 W1, 1 uint32 SAD, 3 KL dot, 4 uint16 SAD (headline).  <KIND, 4, false> matrix output, <KIND, 4, true> k-nearest sweep.
CONTRACT = [
    # the matrix kernels of the integer kinds and of raw fp64 W1 / KL: two waves per SIMD, nothing spilled, no scratch
    (r"pair_stream_kernelILi4ELi4ELb0E", dict(max_regs=256, spills=False)),
    (r"pair_stream_kernelILi0ELi4ELb0E", dict(max_regs=256, spills=False)),
    (r"pair_stream_kernelILi3ELi4ELb0E", dict(max_regs=256, spills=False)),
    (r"pair_stream_kernelILi1ELi4ELb0E", dict(max_regs=256, max_scratch=256)),
    # the k-nearest sweeps carry their candidate epilogue on top of the same tile loop: two waves per SIMD, and (as built) nothing spilled
    (r"pair_stream_kernelILi\dELi4ELb1E", dict(max_regs=256, spills=False)),
    # the one-tile kernels: 128 x 128 tiles run two workgroups per CU, 64 x 64 up to five
    (r"pair_kernelILi\dELb[01]ELi8E", dict(max_regs=256, max_scratch=256)),
    (r"pair_kernelILi\dELb[01]ELi4E", dict(max_regs=128, max_scratch=256)),
]


def find_tool(name):
    cands = [os.environ.get("HSD_" + name.upper().replace("-", "_")), os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin", name), shutil.which(name)]
    for c in cands:
        if c and os.path.isfile(c) and os.access(c, os.X_OK):
            return c
    return None


def code_object_notes(lib):
    """yield (code object name, text of llvm-readelf --notes) for every gfx950 code object bundled in the shared library"""
    objdump, readelf = find_tool("llvm-objdump"), find_tool("llvm-readelf")
    if not objdump or not readelf:
        raise RuntimeError("llvm-objdump / llvm-readelf not found (ROCm LLVM tools)")
    tmp = tempfile.mkdtemp(prefix="hsd_res_")
    try:
        local = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, local)
        subprocess.check_call([objdump, "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" in f:
                yield f, subprocess.check_output([readelf, "--notes", os.path.join(tmp, f)], text=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


FIELDS = ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size",
          "max_flat_workgroup_size")


def parse_kernels(notes_text):
    """[{name, vgpr_count, ...}] from the AMDGPU metadata YAML of one code object (flat scan: every kernel entry begins at `- .` of the
    amdhsa.kernels list; argument lists are nested deeper and carry none of the fields read here)."""
    kernels, cur, in_kernels = [], None, False
    for line in notes_text.splitlines():
        if line.startswith("amdhsa.kernels:"):
            in_kernels = True
            continue
        if in_kernels and re.match(r"^amdhsa\.\w+:", line):
            in_kernels = False
        if not in_kernels:
            continue
        m = re.match(r"^  - \.(\w+):\s*(.*)$", line)
        if m:                                        # first key of a new kernel entry
            cur = {}
            kernels.append(cur)
            key, val = m.group(1), m.group(2)
        else:
            m = re.match(r"^    \.(\w+):\s*(.*)$", line)
            if not m or cur is None:
                continue
            key, val = m.group(1), m.group(2)
        if key == "name":
            cur["name"] = val.strip()
        elif key in FIELDS:
            try:
                cur[key] = int(val.strip())
            except ValueError:
                pass
    return [k for k in kernels if "name" in k]


def check_kernel(k, contract=CONTRACT):
    """list of violations (strings) of the first matching rule; [] when the kernel is within its contract or matches no rule"""
    for pattern, rule in contract:
        if not re.search(pattern, k["name"]):
            continue
        bad = []
        regs = k.get("vgpr_count", 0) + k.get("agpr_count", 0)
        scratch = k.get("private_segment_fixed_size", 0)
        if "max_regs" in rule and regs > rule["max_regs"]:
            bad.append(f"{regs} vector registers (vgpr {k.get('vgpr_count', 0)} + agpr {k.get('agpr_count', 0)}) > {rule['max_regs']}: "
                       f"{VGPR_FILE_PER_SIMD // max(regs, 1)} wave(s) per SIMD instead of {VGPR_FILE_PER_SIMD // rule['max_regs']}")
        if rule.get("spills") is False:
            if k.get("vgpr_spill_count", 0) or k.get("sgpr_spill_count", 0) and scratch:
                bad.append(f"spills: {k.get('vgpr_spill_count', 0)} vector / {k.get('sgpr_spill_count', 0)} scalar registers (designed spill-free)")
            if scratch:
                bad.append(f"{scratch} bytes of scratch per lane (designed scratch-free)")
        if "max_scratch" in rule and scratch > rule["max_scratch"]:
            bad.append(f"{scratch} bytes of scratch per lane > {rule['max_scratch']}")
        if "max_lds" in rule and k.get("group_segment_fixed_size", 0) > rule["max_lds"]:
            bad.append(f"{k['group_segment_fixed_size']} bytes of static LDS > {rule['max_lds']}")
        return bad
    return []


def main(argv):
    args = [a for a in argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "libhsd_amd.so")
    listing = "--list" in argv
    failures, seen, matched = [], 0, 0
    for co, notes in code_object_notes(lib):
        for k in parse_kernels(notes):
            seen += 1
            rule_hit = any(re.search(p, k["name"]) for p, _ in CONTRACT)
            matched += rule_hit
            if listing and (rule_hit or "--all" in argv):
                print(f"{k['name'][:70]:70s} vgpr {k.get('vgpr_count', 0):3d} agpr {k.get('agpr_count', 0):3d} sgpr {k.get('sgpr_count', 0):3d} "
                      f"spill v{k.get('vgpr_spill_count', 0)} s{k.get('sgpr_spill_count', 0)} scratch {k.get('private_segment_fixed_size', 0):4d} "
                      f"lds {k.get('group_segment_fixed_size', 0)}")
            for b in check_kernel(k):
                failures.append(f"{k['name']}: {b}")
    if matched < 8:
        failures.append(f"only {matched} of the all-pairs kernels were found in {seen} kernels: the contract patterns no longer match the symbols")
    if failures:
        print("check_resources: the kernels left their register / scratch / LDS contract:", file=sys.stderr)
        for f in failures:
            print("  " + f, file=sys.stderr)
        return 1
    print(f"check_resources: {matched} all-pairs kernels of {seen} within their register / scratch / LDS contract: ok")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
