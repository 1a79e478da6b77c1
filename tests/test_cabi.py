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
