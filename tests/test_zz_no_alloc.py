"""cgn_lstm_seq_fwd_persistent_noalloc_f32 (include/cgn_hip.h): the persistent LSTM forward launch whose arrival
counters live in a file-scope device array of the library, for callers that cannot allocate. Called through the C-ABI
and held bit for bit to the existing entry point cgn_lstm_seq_fwd_persistent_f32 (same kernels, caller-owned sync_ws),
on two calls in a row, the second with a smaller input (the array keeps the first call's counters)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    import ldrs_amd  # noqa: F401
    from ldrs_amd.hip import _lib
    return _lib, _lib.load()


def _inputs(nseq, B, T, H, seed, dev):
    g = torch.Generator().manual_seed(seed)
    seqs = []
    for i in range(nseq):
        seqs.append(dict(xg=torch.randn(B, T, 4 * H, generator=g).to(dev),
                         xg_extra=(torch.randn(B, 4 * H, generator=g).to(dev) if i == 0 else None),
                         w_hh=(torch.randn(4 * H, H, generator=g) / H ** 0.5).to(dev),
                         reverse=i % 2))
    return seqs


def _prepare(nseq, B, T, H, seqs, dev):
    """Output buffers (poisoned) + the host-side descriptor array: every allocation happens HERE, before the call."""
    L, lib = _lib()
    arr = (L.LstmFwdSeq * nseq)()
    outs = []
    for i, s in enumerate(seqs):
        hbuf = torch.full((B, T + 1, H), float("nan"), device=dev)
        cbuf = torch.full((B, T + 1, H), float("nan"), device=dev)
        gates = torch.full((B, T, 4 * H), float("nan"), device=dev)
        arr[i].xg, arr[i].w_hh = s["xg"].data_ptr(), s["w_hh"].data_ptr()
        arr[i].xg_extra = s["xg_extra"].data_ptr() if s["xg_extra"] is not None else None
        arr[i].hbuf, arr[i].cbuf, arr[i].gates = hbuf.data_ptr(), cbuf.data_ptr(), gates.data_ptr()
        arr[i].reverse, arr[i].init_zero = s["reverse"], 1
        outs.append((hbuf, cbuf, gates))
    return arr, outs


def _call(entry, arr, nseq, B, T, H, sync=None, status=None):
    L, lib = _lib()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if entry == "noalloc":
        return lib.cgn_lstm_seq_fwd_persistent_noalloc_f32(stream, arr, nseq, B, T, H,
                                                           C.c_void_p(status.data_ptr() if status is not None else 0))
    return lib.cgn_lstm_seq_fwd_persistent_f32(stream, arr, nseq, B, T, H, C.c_void_p(sync.data_ptr()),
                                               sync.numel() * 4)


def _run(entry, nseq, B, T, H, seqs, dev, sync=None, status=None, check_no_alloc=False):
    arr, outs = _prepare(nseq, B, T, H, seqs, dev)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    reserved_before = torch.cuda.memory_reserved()
    rc = _call(entry, arr, nseq, B, T, H, sync=sync, status=status)
    torch.cuda.synchronize()
    if check_no_alloc:  # neither the device's free memory nor torch's pool moved across the call
        assert torch.cuda.mem_get_info()[0] == free_before and torch.cuda.memory_reserved() == reserved_before
    return rc, outs


def test_noalloc_entry_matches_existing_entry_twice(dev):
    L, lib = _lib()
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    # second call smaller in every dimension: fewer sequences, batch tiles, timesteps and hidden units
    for nseq, B, T, H, seed in ((2, 64, 21, 512, 1), (1, 20, 7, 256, 2)):
        need = C.c_int64(0)
        assert lib.cgn_lstm_persistent_query(nseq, B, T, H, C.byref(need)) == 0 and need.value > 0
        sync = torch.zeros(need.value // 4 + 64, dtype=torch.int32, device=dev)
        seqs = _inputs(nseq, B, T, H, seed, dev)
        rc_ref, ref = _run("ws", nseq, B, T, H, seqs, dev, sync=sync)
        if seed == 1:  # the first call on a device resolves the array's address (module load): not part of the claim
            _run("noalloc", nseq, B, T, H, seqs, dev, status=status)
        rc_new, new = _run("noalloc", nseq, B, T, H, seqs, dev, status=status, check_no_alloc=True)
        assert rc_ref == 0 and rc_new == 0
        assert int(sync[-32].item()) == 0 and int(status.item()) == 0  # no bounded wait ran out in either launch
        for (h0, c0, g0), (h1, c1, g1) in zip(ref, new):
            for a, b, nm in ((h0, h1, "hbuf"), (c0, c1, "cbuf"), (g0, g1, "gates")):
                assert not torch.isnan(b).any(), nm
                assert torch.equal(a, b), (nm, nseq, B, T, H)  # same kernel, same summation order: bit-exact


def test_noalloc_entry_under_graph_capture(dev):
    """What the entry point is for: issued while a stream is being captured (no allocation, no memset, ONE kernel node),
    then replayed twice; each replay reproduces the caller-workspace entry bit for bit (the counters are back at zero
    after every launch). The entry point must have been called once on the device OUTSIDE capture before (that call
    resolves the array's address), as include/cgn_hip.h says."""
    L, lib = _lib()
    nseq, B, T, H = 2, 48, 9, 256
    need = C.c_int64(0)
    assert lib.cgn_lstm_persistent_query(nseq, B, T, H, C.byref(need)) == 0 and need.value > 0
    sync = torch.zeros(need.value // 4 + 64, dtype=torch.int32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    seqs = _inputs(nseq, B, T, H, 5, dev)
    rc_ref, ref = _run("ws", nseq, B, T, H, seqs, dev, sync=sync)
    rc0, _ = _run("noalloc", nseq, B, T, H, seqs, dev, status=status)  # outside capture first
    assert rc_ref == 0 and rc0 == 0
    arr, outs = _prepare(nseq, B, T, H, seqs, dev)
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream(dev)
    st.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(st):
        g.capture_begin()
        rc = _call("noalloc", arr, nseq, B, T, H, status=status)
        g.capture_end()
    assert rc == 0
    torch.cuda.current_stream(dev).wait_stream(st)
    for _ in range(2):
        for t3 in outs:
            for t_ in t3:
                t_.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        assert int(status.item()) == 0
        for (h0, c0, g0), (h1, c1, g1) in zip(ref, outs):
            assert torch.equal(h0, h1) and torch.equal(c0, c1) and torch.equal(g0, g1)


def test_noalloc_entry_rejects_what_it_cannot_hold(dev):
    L, lib = _lib()
    seqs = _inputs(1, 4112, 2, 16, 3, dev)
    rc, _ = _run("noalloc", 1, 4112, 2, 16, seqs, dev)  # 257 batch tiles: beyond the built-in scratch (and the CUs)
    assert rc == L.CGN_ERR_INVALID_ARG
    seqs = _inputs(1, 16, 3, 30, 3, dev)
    rc, _ = _run("noalloc", 1, 16, 3, 30, seqs, dev)  # H % 16 != 0
    assert rc == L.CGN_ERR_INVALID_ARG
