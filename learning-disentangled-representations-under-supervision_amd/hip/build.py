"""Builds csrc/*.hip into lib/libcgn_hip.so for gfx950 with hipcc (no GPU needed: hipcc cross-compiles).

The shared library is a plain C-ABI object (include/cgn_hip.h); it links only against the HIP runtime.
"""
import hashlib
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
OBJ_DIR = os.path.join(PKG_DIR, "build")
LIB_PATH = os.path.join(LIB_DIR, "libcgn_hip.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


PUBLIC_HEADER = os.path.join(os.path.dirname(PKG_DIR), "include", "cgn_hip.h")
GEMM_SECTION_END = "Temperature softmax + NLL"  # first words of the section that follows the GEMM declarations


def _code_only(text):
    """Source text without comments and blank runs: documentation edits do not rebuild anything."""
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return re.sub(r"\s+", " ", text)


def _fingerprint(src):
    """Hash of everything a translation unit is compiled from: its own text, the local headers it includes, the public
    header (code only; the three GEMM units see only its GEMM section - they take ~7 minutes each to compile and use
    nothing else of it) and the compiler flags."""
    h = hashlib.sha256(" ".join(FLAGS).encode())
    text = open(os.path.join(CSRC, src)).read()
    h.update(text.encode())
    for inc in sorted(set(re.findall(r'#include\s+"([\w.]+)"', text)) | {"common.h"}):
        path = os.path.join(CSRC, inc)
        if os.path.exists(path):
            h.update(open(path).read().encode())
    pub = open(PUBLIC_HEADER).read()
    if src.startswith("gemm"):
        # (the marker sits inside a comment: cut the raw text there, then drop the comments)
        pub = pub[:pub.index(GEMM_SECTION_END)]
    h.update(_code_only(pub).encode())
    return h.hexdigest()


def _compile(src, force):
    obj = os.path.join(OBJ_DIR, src[:-4] + ".o")
    dep = obj + ".dep"
    path = os.path.join(CSRC, src)
    fp = _fingerprint(src)
    if not force and os.path.exists(obj) and os.path.exists(dep) and open(dep).read().strip() == fp:
        return obj, False
    cmd = [HIPCC] + FLAGS + ["-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(dep, "w") as f:
        f.write(fp + "\n")
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(LIB_DIR, exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in results]
    rebuilt = any(c for _, c in results)
    if rebuilt or not os.path.exists(LIB_PATH):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", LIB_PATH)
    elif verbose:
        print("up to date:", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
