"""CPU checks of the drop-in boundary: include/cgn_hip.h <-> libcgn_hip.so <-> the ctypes binding table.

No compute call is made here (no GPU in the CPU tier): the library only has to load, export every declared symbol,
answer cgn_abi_version(), and agree with the header on struct layouts (checked by compiling the header with gcc).
"""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "cgn_hip.h")

import ldrs_amd  # noqa: E402,F401  (aliases the hyphenated package directory)
from ldrs_amd.hip import _lib  # noqa: E402


def _declarations():
    """{name: n_args} of every `int cgn_*(...)` prototype in the header."""
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    out = {}
    for m in re.finditer(r"\bint\s+(cgn_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    return out


def test_header_and_binding_table_agree():
    decl = _declarations()
    assert len(decl) >= 30
    assert set(decl) == set(_lib.PROTOTYPES), (
        "header-only: %s, binding-only: %s" % (sorted(set(decl) - set(_lib.PROTOTYPES)),
                                               sorted(set(_lib.PROTOTYPES) - set(decl))))
    for name, n in decl.items():
        assert len(_lib.PROTOTYPES[name][1]) == n, "%s: header has %d args, binding %d" % (
            name, n, len(_lib.PROTOTYPES[name][1]))


def test_library_loads_and_exports_every_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build first: python __graft_entry__.py build"
    lib = _lib.load()
    for name in _declarations():
        assert hasattr(lib, name), "libcgn_hip.so does not export %s" % name
    assert lib.cgn_abi_version() == _source_abi_version()


def _source_abi_version():
    src = open(os.path.join(_lib.PKG_DIR, "csrc", "abi.hip")).read()
    return int(re.search(r"cgn_abi_version\(void\)\s*\{\s*return\s+(\d+)\s*;", src).group(1))


def _header_struct_fields(cname):
    """Field names of `typedef struct <cname> { ... }` in the header, in declaration order."""
    text = re.sub(r"/\*.*?\*/", " ", open(HEADER).read(), flags=re.S)
    body = re.search(r"typedef\s+struct\s+%s\s*\{(.*?)\}\s*%s\s*;" % (cname, cname), text, flags=re.S).group(1)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        first, *rest = decl.split(",")
        names.append(re.search(r"(\w+)\s*(?:\[\w*\])?\s*$", first).group(1))
        names += [re.search(r"(\w+)", r_).group(1) for r_ in rest]
    return names


def test_integration_md_shows_the_shipped_abi():
    """INTEGRATION.md §B is what a maintainer of the reference copies: its version assert and its cgn_gemm_desc /
    cgn_axis field lists must be the header's (round 2 shipped a document one ABI behind, 16 bytes short)."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = doc[doc.index("<!-- ABI:BEGIN"):doc.index("<!-- ABI:END -->")]
    assert int(re.search(r"cgn_abi_version\(\) == (\d+)", block).group(1)) == _source_abi_version()
    assert "cgn_abi_version() ==" not in doc.replace(block, ""), "a second, hand-written version assert"
    for pyname, cname, cls in (("Axis", "cgn_axis", _lib.Axis), ("GemmDesc", "cgn_gemm_desc", _lib.GemmDesc)):
        m = re.search(r"class %s\(C\.Structure\):.*?_fields_ = \[(.*?)\]\n" % pyname, block, flags=re.S)
        fields = re.findall(r'\("(\w+)",\s*([\w.]+)\)', m.group(1))
        assert [f for f, _ in fields] == _header_struct_fields(cname), pyname
        assert [f for f, _ in fields] == [f for f, _ in cls._fields_], pyname
        ns = {"C": C, "Axis": _lib.Axis}
        doc_cls = type(pyname, (C.Structure,), {"_fields_": [(f, eval(t, ns)) for f, t in fields]})
        assert C.sizeof(doc_cls) == C.sizeof(cls), pyname
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_integration_abi.py"), "--check"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(ROOT, "no_such_dir", "libcgn_hip.so"))
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        _lib.load()


def test_error_codes_raise():
    _lib.check(_lib.CGN_OK, "x")
    with pytest.raises(RuntimeError, match="invalid argument"):
        _lib.check(_lib.CGN_ERR_INVALID_ARG, "x")
    with pytest.raises(RuntimeError, match="workspace"):
        _lib.check(_lib.CGN_ERR_WORKSPACE, "x")
    with pytest.raises(RuntimeError, match="HIP error 700"):
        _lib.check(_lib.CGN_ERR_HIP_BASE + 700, "x")


STRUCTS = {"cgn_axis": _lib.Axis, "cgn_gemm_desc": _lib.GemmDesc, "cgn_lstm_fwd_seq": _lib.LstmFwdSeq,
           "cgn_lstm_bwd_seq": _lib.LstmBwdSeq, "cgn_conv_widths": _lib.ConvWidths,
           "cgn_code_head_fwd": _lib.CodeHeadFwd, "cgn_code_head_bwd": _lib.CodeHeadBwd,
           "cgn_lstm2_fwd": _lib.Lstm2Fwd, "cgn_lstm2_bwd": _lib.Lstm2Bwd}


def test_struct_layouts_match_header(tmp_path):
    """The header is plain C: compile it with gcc, print sizeof/offsetof of every struct field, compare with ctypes."""
    lines = ['#include <stdio.h>', '#include "cgn_hip.h"', "int main(void){"]
    for cname, cls in STRUCTS.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o",
                           str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for cname, cls in STRUCTS.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got["%s.%s" % (cname, fname)]) == getattr(cls, fname).offset, "%s.%s" % (cname, fname)


def test_header_constants_match_binding():
    text = open(HEADER).read()
    consts = dict(re.findall(r"#define\s+(CGN_\w+)\s+(\d+)", text))
    assert int(consts["CGN_OK"]) == _lib.CGN_OK
    assert int(consts["CGN_ERR_INVALID_ARG"]) == _lib.CGN_ERR_INVALID_ARG
    assert int(consts["CGN_ERR_WORKSPACE"]) == _lib.CGN_ERR_WORKSPACE
    assert int(consts["CGN_ERR_HIP_BASE"]) == _lib.CGN_ERR_HIP_BASE
    assert int(consts["CGN_ATRANS_EXP"]) == _lib.ATRANS_EXP
    assert int(consts["CGN_EPI_SOFTMAX_BWD"]) == _lib.EPI_SOFTMAX_BWD
    assert int(consts["CGN_ACT_RELU"]) == _lib.ACT_RELU and int(consts["CGN_ACT_TANH"]) == _lib.ACT_TANH


def test_ops_refuse_cpu_tensors():
    """Host-side validation: the wrappers reject non-HIP tensors before anything reaches a kernel."""
    import torch
    from ldrs_amd.hip import ops
    a = torch.zeros(4, 4)
    with pytest.raises(TypeError, match="HIP"):
        ops.fill(a, 1.0)
