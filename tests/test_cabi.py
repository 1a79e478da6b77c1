"""The C-ABI shared library loads and exports every symbol include/rsvld_hip.h declares (no compute)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "rsvld_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rsvld_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import ctypes
    from rsvld_amd import _lib
    names = _declared()
    assert len(names) >= 15
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"
    lib = _lib.load()
    for n in names:
        assert isinstance(getattr(lib, n), ctypes._CFuncPtr)
    assert b"gfx950" in lib.rsvld_version()


def test_conv_desc_matches_header_layout():
    """Field order / count of the ctypes struct follows the C struct declaration."""
    from rsvld_amd import _lib
    src = open(os.path.join(ROOT, "include", "rsvld_hip.h")).read()
    start = src.index("typedef struct rsvld_conv_desc {") + len("typedef struct rsvld_conv_desc {")
    body = re.sub(r"/\*.*?\*/", "", src[start:src.index("} rsvld_conv_desc;")], flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for nm in decl.split(","):
            fields.append(nm.strip().split()[-1].lstrip("*"))
    assert fields == [f[0] for f in _lib.ConvDesc._fields_]


def test_constants_match_header():
    """The dtype / activation / tune constants of the ctypes layer are the header's #defines (a flag added on one side only would
    silently select another kernel)."""
    from rsvld_amd import _lib
    src = open(os.path.join(ROOT, "include", "rsvld_hip.h")).read()
    defs = {}
    for name, expr in re.findall(r"^#define\s+(RSVLD_[A-Z0-9_]+)\s+(\(?-?[0-9]+(?:\s*<<\s*[0-9]+)?\)?)", src, flags=re.M):
        defs[name] = eval(expr)                      # integer literals and shifts only (the regular expression admits nothing else)
    want = {"RSVLD_F16": _lib.F16, "RSVLD_BF16": _lib.BF16, "RSVLD_F32": _lib.F32, "RSVLD_SPLIT": _lib.SPLIT,
            "RSVLD_F16W2": _lib.F16W2, "RSVLD_F16W1": _lib.F16W1, "RSVLD_F16Q8": _lib.F16Q8,
            "RSVLD_ACT_NONE": _lib.ACT_NONE, "RSVLD_ACT_SILU": _lib.ACT_SILU, "RSVLD_ACT_GEGLU": _lib.ACT_GEGLU,
            "RSVLD_TUNE_STAGES_SHIFT": _lib.TUNE_STAGES_SHIFT, "RSVLD_TUNE_NO_KSPLIT": _lib.TUNE_NO_KSPLIT,
            "RSVLD_TUNE_REG_STAGING": _lib.TUNE_REG_STAGING, "RSVLD_TUNE_HALO_NW4": _lib.TUNE_HALO_NW4,
            "RSVLD_TUNE_HALO_NW8": _lib.TUNE_HALO_NW8, "RSVLD_TUNE_NO_GEMM256": _lib.TUNE_NO_GEMM256,
            "RSVLD_TUNE_GEMM_ONE_TILE": _lib.TUNE_GEMM_ONE_TILE, "RSVLD_TUNE_F32_SPLIT": _lib.TUNE_F32_SPLIT}
    for k, v in want.items():
        assert defs.get(k) == v, (k, defs.get(k), v)
    flags = [v for k, v in defs.items() if k.startswith("RSVLD_TUNE_") and k not in ("RSVLD_TUNE_TILE_MASK", "RSVLD_TUNE_STAGES_SHIFT")]
    assert len(set(flags)) == len(flags) and all(f & (f - 1) == 0 and f > 63 for f in flags), "tune flags must be distinct single bits above the tile / stage fields"
