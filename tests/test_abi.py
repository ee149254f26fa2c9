"""CPU: the C-ABI library loads and exports every symbol include/hsd.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

from conftest import REPO


def _declared():
    text = open(os.path.join(REPO, "include", "hsd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hsd_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    from hsd_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/hsd.h but not exported"
    # the binding table covers the header exactly
    assert sorted(_lib.SIGNATURES) == names
    # ... and the dynamic symbol table holds nothing but the library's own hsd_* names (csrc/libhsd_amd.map): no C++ internals, no kernel
    # stubs, none of the rocprim templates tkorder.hip instantiates -- nothing a host application's own copies could collide with
    import shutil
    import subprocess
    nm = shutil.which("nm")
    if nm:
        out = subprocess.run([nm, "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
        exported = sorted(ln.split()[-1] for ln in out.splitlines() if ln.strip())
        assert exported and all(e.startswith("hsd_") for e in exported), [e for e in exported if not e.startswith("hsd_")][:5]
        assert set(names) <= set(exported) and len(exported) <= len(names) + 4, sorted(set(exported) - set(names))


def test_loader_sets_prototypes_and_version():
    from hsd_amd import _lib
    lib = _lib.load()
    assert lib.hsd_abi_version() == 2 == _lib.ABI_VERSION
    assert lib.hsd_strerror(-5) == b"Distribution can't be empty."
    assert lib.hsd_strerror(-8) == b"The sum of probability distribution must be 1.0."
    assert lib.hsd_strerror(_lib.ERR_NOMEM) == b"out of host memory" and b"exception" in lib.hsd_strerror(_lib.ERR_INTERNAL)


def test_every_entry_point_stops_exceptions_at_the_boundary():
    """Every int-returning extern "C" definition of the library is a function-try-block closed by HSD_ABI_GUARD: no C++ exception (bad_alloc
    from a std::vector, system_error from a thread that cannot be created) reaches a ctypes / cgo / JNI caller as std::terminate."""
    csrc = os.path.join(REPO, "hsd_amd", "csrc")
    n_defs = 0
    for f in sorted(os.listdir(csrc)):
        if not f.endswith(".hip"):
            continue
        text = open(os.path.join(csrc, f), encoding="utf-8").read()
        for m in re.finditer(r'^extern "C" int (hsd_[a-z0-9_]+)\(([^;{]*?)\)\s*(try)?\s*\{', text, flags=re.M | re.S):
            if m.group(1) == "hsd_abi_version":
                continue
            n_defs += 1
            assert m.group(3) == "try", f"{f}: {m.group(1)} is not a function-try-block"
        assert text.count(") try {") == text.count("} HSD_ABI_GUARD("), f
    assert n_defs >= 40


def test_environment_hooks_are_one_table_and_documented():
    """csrc/hooks.hpp is the only place the library reads the environment (plus the table printer), and DESIGN.md section 9 names every hook."""
    csrc = os.path.join(REPO, "hsd_amd", "csrc")
    hooks = re.findall(r'^\s*H_([A-Z0-9_]+),', open(os.path.join(csrc, "hooks.hpp"), encoding="utf-8").read(), flags=re.M)
    assert len(hooks) >= 30
    table = open(os.path.join(csrc, "context.hip"), encoding="utf-8").read()
    design = open(os.path.join(REPO, "DESIGN.md"), encoding="utf-8").read()
    assert re.findall(r'\{"HSD_([A-Z0-9_]+)", "', table) == hooks, "hsd_hook_table must list the hooks in the order of the enum (it is indexed by it)"
    for h in hooks:
        assert f'{{"HSD_{h}"' in table, f"HSD_{h} missing from hsd_hook_table"
        assert f"HSD_{h}" in design, f"HSD_{h} is not documented in DESIGN.md"
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".hpp")):
            text = open(os.path.join(csrc, f), encoding="utf-8").read()
            stray = [m for m in re.findall(r'getenv\("([A-Z_0-9]+)"\)', text) if m != "HSD_HOOKS_LIST"]
            assert not stray, (f, stray)


def test_ablation_hooks_are_not_in_the_shipped_build():
    """HSD_DEBUG / HSD_TOPK_DEBUG make a call return wrong results or fail on purpose (timing ablations): only a -DHSD_DEBUG_HOOKS build (`make debug`,
    loaded through HSD_LIB) reads them; the default build's flags do not define it (tests/test_gpu_tools.py checks the behaviour on the GPU)."""
    csrc = os.path.join(REPO, "hsd_amd", "csrc")
    hooks = open(os.path.join(csrc, "hooks.hpp"), encoding="utf-8").read()
    assert "hsd_hook_is_ablation" in hooks and "#ifndef HSD_DEBUG_HOOKS" in hooks and "h == H_DEBUG || h == H_TOPK_DEBUG" in hooks
    mk = open(os.path.join(csrc, "Makefile"), encoding="utf-8").read()
    flags = [ln for ln in mk.splitlines() if ln.startswith("CXXFLAGS")]
    assert len(flags) == 1 and "HSD_DEBUG_HOOKS" not in flags[0] and "debug:" in mk and "-DHSD_DEBUG_HOOKS" in mk


def test_no_oracle_import_in_product():
    """The product path must never route through the CPU oracle."""
    pkg = os.path.join(REPO, "hsd_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(root, f), encoding="utf-8").read()
                assert "oracle" not in src.replace("no CPU fallback", ""), f"{f} mentions the oracle"


def _build_c_client(tmp_path):
    import subprocess
    exe = str(tmp_path / "abi_client")
    libdir = os.path.join(REPO, "hsd_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"),
                           os.path.join(REPO, "tests", "c", "abi_client.c"), "-o", exe, "-L", libdir, "-lhsd_amd",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_plain_c_client_links_and_runs(tmp_path):
    """include/hsd.h is a C header: a C99 client compiled with gcc links against the library and gets the
    documented status codes (what a cgo / JNI binding relies on).  No GPU call beyond hsd_create(0, NULL)."""
    import subprocess
    exe = _build_c_client(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "abi ok" in out.stdout
