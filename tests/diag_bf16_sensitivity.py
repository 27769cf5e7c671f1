"""Diagnostic (test infrastructure, run by hand: python tests/diag_bf16_sensitivity.py on a GPU box; not collected by
pytest): which products of the step carry the bf16 mode's gradient error? Runs the 'mid' parity case in bf16 mode
with the products issued from chosen host functions forced back to fp32 and prints the worst gradient errors."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import ldrs_amd  # noqa
from ldrs_amd.hip import ops
import test_step_gpu as T

dev = torch.device("cuda", 0)
real = ops.gemm
for force in [(), ("highway_fwd",), ("highway_bwd",), ("highway_fwd", "highway_bwd"), ("encoder_fwd", "_encode"),
              ("ALLFWD",)]:
    def patched(*a, **kw):
        fn = sys._getframe(1).f_code.co_name
        if fn in force or ("ALLFWD" in force and "bwd" not in fn and "backward" not in fn):
            kw["compute"] = "f32"
        return real(*a, **kw)
    ops.gemm = patched
    ops.set_compute("bf16")
    import io, contextlib
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            T._step_parity(dev, "mid", 16, False, 1e-2)
    except AssertionError as e:
        print("assert", str(e)[:200])
    rows = [l.split() for l in buf.getvalue().splitlines() if l.startswith("bf16")]
    rows = [(float(r[-1]), r[-2]) for r in rows if r[3] == "0"]
    rows.sort()
    print("forced fp32:", force, "callers seen")
    for v, n in rows[-5:]:
        print("   %-50s %.2e" % (n, v))
    print("   non-TDNN max: %.2e" % max(v for v, n in rows if "TDNN" not in n and "char" not in n), flush=True)
ops.gemm = real
