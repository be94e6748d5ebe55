"""The C-ABI library loads without a GPU and exports every symbol that
include/fea_unet.h declares; the ctypes table mirrors the header."""
import os
import re

import pytest

from fea_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "fea_unet.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|size_t|const char\*)\s+(fea_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args == "void" else len([a for a in args.split(",") if a.strip()])
    return out


def test_library_exports_every_declared_symbol():
    if not os.path.exists(L.LIB_PATH):
        pytest.fail(f"{L.LIB_PATH} is not built (run __graft_entry__.build())")
    lib = L.load()
    decl = header_functions()
    assert len(decl) >= 30
    for name in decl:
        assert hasattr(lib, name), name
    assert lib.fea_version() >= 100


def test_ctypes_table_matches_header():
    decl = header_functions()
    assert set(decl) == set(L.SIGNATURES), set(decl) ^ set(L.SIGNATURES)
    for name, nargs in decl.items():
        assert len(L.SIGNATURES[name][1]) == nargs, name


def test_no_cpu_fallback_in_product():
    """Nothing under fea_amd/ may import the oracle."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "fea_amd")):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), os.path.join(dirpath, f)
                assert "import_module(\"oracle" not in text and "__import__(\"oracle" not in text


def test_asm_hazard_checker_finds_an_early_use_of_a_pending_load():
    """scripts/check_asm_hazards.py (run by the Makefile on head_loss.hip, whose global loads are inline asm retired by
    counted waits): a use of a load's destination before the vmcnt that retires it, and any scratch access, are findings."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import check_asm_hazards as C
    clean = """
_Z4kernv:
	global_load_dwordx4 v[10:13], v[2:3], off
	global_load_dwordx4 v[14:17], v[2:3], off offset:64
	v_add_f32_e32 v1, v4, v5
	s_waitcnt vmcnt(1)
	v_add_f32_e32 v1, v10, v11
	s_waitcnt vmcnt(0)
	v_add_f32_e32 v1, v14, v15
	s_endpgm
"""
    early = clean.replace("s_waitcnt vmcnt(1)\n\tv_add_f32_e32 v1, v10, v11", "s_waitcnt vmcnt(1)\n\tv_add_f32_e32 v1, v10, v16")
    spill = clean.replace("\tv_add_f32_e32 v1, v4, v5", "\tscratch_store_dword off, v4, off")
    assert C.scan(clean, "kern") == 0
    assert C.scan(early, "kern") == 1
    assert C.scan(spill, "kern") == 1
    assert C.scan(clean, "no_such_kernel") == 2
