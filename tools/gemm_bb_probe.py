"""Diagnostic: the bf16 x bf16 product (csrc/bb_gemm.hip) — correctness against an fp64 product of the bf16-rounded
operands on awkward shapes, then time beside the kernel that rounds fp32 operands at staging (ops.gemm, bf16 compute)
on the step's shapes at BASELINE configs[1] and configs[3].
usage: gemm_bb_probe.py [out.txt] [--quick]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ldrs_amd  # noqa
from ldrs_amd.hip import ops
from ldrs_amd.hip._lib import Axis

dev = torch.device("cuda", 0)
torch.manual_seed(0)
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


def timed(fn, reps=10):
    us = float("inf")
    for _ in range(3):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = min(us, 1e3 * e0.elapsed_time(e1) / reps)
    return us


def check(M, N, K, tile=0, splitk=0, c16=False, bias=False, beta=0.0, alpha=1.0):
    a = torch.randn(M, K, device=dev)
    b = torch.randn(N, K, device=dev)
    A = ops.cast_bf16(a, Axis(K, 0, 0), M, K)
    B = ops.cast_bf16(b, Axis(K, 0, 0), N, K)
    assert torch.equal(A[:, :K], a.to(torch.bfloat16)) and (A[:, K:] == 0).all()
    out = torch.randn(M, N, device=dev)
    c0 = out.clone()
    if c16:
        out = out.to(torch.bfloat16)
    b0 = torch.randn(N, device=dev) if bias else None
    b1 = torch.randn(N, device=dev) if bias else None
    ops.gemm_bb(A, B, out, alpha=alpha, beta=beta, bias0=b0, bias1=b1, force_tile=tile, force_splitk=splitk)
    ref = alpha * (A[:, :K].double() @ B[:, :K].double().t())
    if bias:
        ref = ref + b0.double() + b1.double()
    if beta:
        ref = ref + beta * c0.double()
    scale = (A[:, :K].double().abs() @ B[:, :K].double().abs().t()) + 1.0
    err = ((out.double() - ref).abs() / scale).max().item()
    bar = 6e-3 if c16 else 2e-6
    say("check M=%5d N=%5d K=%5d tile=%3d sk=%2d c16=%d bias=%d beta=%.1f: worst %.2e %s"
        % (M, N, K, tile, splitk, c16, bias, beta, err, "ok" if err < bar else "FAIL"))
    return err < bar


def check_cast():
    ok = True
    src = torch.randn(37, 21, 52, device=dev)  # [B, T+1, H] history: rows (b, t) with t < 20
    ax = Axis(52, 20, 21 * 52)
    rows = src[:, :20].reshape(-1, 52)
    d = ops.cast_bf16(src, ax, 37 * 20, 52)
    ok &= torch.equal(d[:, :52], rows.to(torch.bfloat16)) and bool((d[:, 52:] == 0).all())
    dt = ops.cast_bf16(src, ax, 37 * 20, 52, transpose=True)
    ok &= torch.equal(dt[:, :740], rows.t().to(torch.bfloat16)) and bool((dt[:, 740:] == 0).all())
    m = torch.randn(301, 777, device=dev)
    v = m[:, 1:]  # unaligned rows: scalar loads
    d = ops.cast_bf16(v, Axis(777, 0, 0), 301, 776)
    ok &= torch.equal(d[:, :776], v.to(torch.bfloat16))
    dt = ops.cast_bf16(v, Axis(777, 0, 0), 301, 776, transpose=True)
    ok &= torch.equal(dt[:, :301], v.t().to(torch.bfloat16)) and bool((dt[:, 301:] == 0).all())
    say("casts (history axis, transposed, unaligned): %s" % ("ok" if ok else "FAIL"))
    return ok


def speed(name, M, N, K, layout, c16=False, a16=False):
    """layout: 'nt' A [M,K] k-contiguous, B [N,K] k-contiguous; 'nn' B [K,N]; 'tn' A [K,M], B [K,N]"""
    a = torch.randn(M, K, device=dev) if layout[0] == "n" else torch.randn(K, M, device=dev).t()
    if a16:
        a = a.to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).t() if layout[1] == "t" else torch.randn(K, N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16 if c16 else torch.float32)
    prev = ops.set_compute("bf16")
    res = []
    for bb in (False, True):
        ops.BB_GEMM = bb
        fn = lambda: ops.gemm(a, b, out)
        fn()
        torch.cuda.synchronize()
        res.append((timed(fn), out.float().clone()))
    ops.BB_GEMM = True
    ops.set_compute(prev)
    # the product alone (operands cast beforehand)
    A = a if a16 else ops.cast_bf16(a.contiguous(), Axis(K, 0, 0), M, K) if layout[0] == "n" else \
        ops.cast_bf16(a.t(), Axis(M, 0, 0), K, M, transpose=True)
    Bc = ops.cast_bf16(b.t().contiguous(), Axis(K, 0, 0), N, K)
    t_alone = timed(lambda: ops.gemm_bb(A, Bc, out, K=A.shape[1]))
    diff = ((res[0][1] - res[1][1]).norm() / res[0][1].norm()).item()
    fl = 2.0 * M * N * K / 1e6
    say("%-26s %s M=%6d N=%6d K=%6d | staged-round %8.1f us %6.0f TF/s | casts + bb %8.1f us %6.0f TF/s | bb alone %8.1f us "
        "%6.0f TF/s | x%.2f | rel diff %.1e" % (name, layout, M, N, K, res[0][0], fl / res[0][0], res[1][0], fl / res[1][0],
                                                t_alone, fl / t_alone, res[0][0] / res[1][0], diff))


def main():
    quick = "--quick" in sys.argv
    ok = check_cast()
    for M, N, K, kw in [
        (256, 256, 64, {}), (256, 256, 96, {}), (300, 520, 328, {}), (513, 257, 1000, dict(bias=True, beta=0.5, alpha=0.7)),
        (1000, 300, 2048, dict(splitk=3)), (640, 900, 4096, dict(tile=256, splitk=4, bias=True, beta=1.0)),
        (128, 128, 64, dict(tile=128)), (200, 1000, 520, dict(tile=128, c16=True, bias=True)),
        (777, 1032, 512, dict(tile=256, c16=True, bias=True)), (1344, 2048, 512, dict(tile=128)),
        (1344, 2048, 512, dict(tile=256)), (260, 264, 8, dict(tile=256)), (130, 3000, 40, {}),
    ]:
        ok &= check(M, N, K, **kw)
    say("correctness: %s" % ("all ok" if ok else "FAILURES"))
    if not ok:
        return 1
    speed("square 4096", 4096, 4096, 4096, "nt")
    speed("square 8192", 8192, 8192, 8192, "nt")
    # configs[1]: NT = 1344, V = 20000, H = 512, E = 300
    speed("cfg2 projection", 1344, 20000, 512, "nt", c16=True)
    speed("cfg2 dH1", 1344, 512, 20000, "nn", a16=True)
    speed("cfg2 dW_fc", 20000, 512, 1344, "nn", a16=True)
    speed("cfg2 LSTM input", 1344, 2048, 512, "nt")
    if not quick:
        # configs[3]: NT = 10496, V = 50000, H = 1024
        speed("cfg4 projection", 10496, 50000, 1024, "nt", c16=True)
        speed("cfg4 dH1", 10496, 1024, 50000, "nn", a16=True)
        speed("cfg4 dW_fc", 50000, 1024, 10496, "nn", a16=True)
        speed("cfg4 LSTM input", 10496, 4096, 1024, "nt")
        speed("cfg4 LSTM input E", 10496, 4096, 300, "nt")
        speed("cfg4 LSTM dX", 10496, 1024, 4096, "nn")
        speed("cfg4 LSTM dW", 4096, 1024, 10240, "tn")
    return 0


if __name__ == "__main__":
    rc = main()
    for a in sys.argv[1:]:
        if not a.startswith("--"):
            with open(a, "w") as f:
                f.write("\n".join(lines) + "\n")
    sys.exit(rc)
