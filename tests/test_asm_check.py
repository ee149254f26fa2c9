"""CPU: the build-time checker of the hand-issued instructions (hsd_amd/csrc/check_m0.py) on synthetic disassembly.

The checker is the only guard of two hazards hipcc does not pad for inline asm; these cases pin what it must flag: a scalar
address register written by ANY vector-ALU instruction (not only the lane reads) within five wait states, also when the write
reaches the memory instruction through a branch, and a branch that enters an M0 sequence."""
import importlib.util
import os

from conftest import REPO


def _mod():
    spec = importlib.util.spec_from_file_location("check_m0", os.path.join(REPO, "hsd_amd", "csrc", "check_m0.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _listing(lines, name="_Z18pair_stream_kernelILi4ELi4ELb0EEv8PairArgs", base=0x1000):
    out = [f"{base:016x} <{name}>:"]
    addr = base
    for body, target in lines:
        tail = f" <{name}+0x{target - base:x}>" if target is not None else ""
        out.append(f"\t{body:<58} // {addr:012X}: 00000000{tail}")
        addr += 4
    return "\n".join(out) + "\n"


def _hazards(m, lines):
    fn = m.parse(_listing(lines))[0]
    fn.build()
    bad = []
    for k, x in enumerate(fn.ins):
        if x.startswith("global_store_dwordx4"):
            m.hazard_walk(fn, k, m.sregs("s[4:5]"), x, bad)
    return fn, bad


def test_any_valu_write_of_the_address_is_flagged():
    m = _mod()
    store = ("global_store_dwordx4 v1, v[2:5], s[4:5] nt", None)
    for writer in ("v_readfirstlane_b32 s4, v0", "v_readlane_b32 s5, v9, 3", "v_cmp_lt_u32_e64 s[4:5], v1, v2", "v_add_co_u32_e64 v1, s[4:5], v2, v3",
                   "v_mad_u64_u32 v[0:1], s[4:5], v2, v3, v[4:5]"):
        _, bad = _hazards(m, [(writer, None), ("s_nop 2", None), store, ("s_nop 1", None)])
        assert bad, writer                                       # 3 wait states < 5
        _, bad = _hazards(m, [(writer, None), ("s_nop 4", None), store, ("s_nop 1", None)])
        assert not bad, (writer, bad)
    # a source operand is not a write
    _, bad = _hazards(m, [("v_add_u32_e32 v1, s4, v2", None), store, ("s_nop 1", None)])
    assert not bad


def test_hazard_through_a_branch():
    m = _mod()
    base = 0x1000
    # 0: writer; 1: s_branch -> 4; 2,3: padded path; 4: store (target of 1, three pads after 2's fall-through path only)
    lines = [("v_readfirstlane_b32 s4, v0", None), ("s_branch 2", base + 4 * 4), ("s_nop 4", None), ("s_nop 0", None),
             ("global_store_dwordx4 v1, v[2:5], s[4:5] nt", None), ("s_nop 1", None)]
    fn, bad = _hazards(m, lines)
    assert fn.is_target[4]
    assert bad and "needs 5" in bad[0]                            # the flat list would have seen six wait states in between


def test_check_reports_missing_objdump(monkeypatch, tmp_path):
    m = _mod()
    monkeypatch.setenv("HSD_OBJDUMP", str(tmp_path / "nope"))
    monkeypatch.setenv("ROCM_PATH", str(tmp_path))
    monkeypatch.setenv("PATH", str(tmp_path))
    assert m.find_objdump() is None
