"""The oracle's Selections aid (oracle/cgn_oracle.py): reporting gate / winner choices must not change the oracle, a
differing choice is adopted only where the oracle's own candidates are tied, and is counted either way."""
import torch

from oracle import cgn_oracle as O

KW = dict(max_word_len=7, max_seq_len=5, word_vocab_size=50, char_vocab_size=20, word_embed_size=12,
          kernels=((1, 3), (2, 5), (3, 4)), encoder_rnn_size=16, latent_variable_size=6, decoder_rnn_size=20,
          sentiment_kernel_size=(3, 4, 5), sentiment_kernel_num=6)


def _run(select):
    ocfg = O.OracleConfig(**KW)
    torch.manual_seed(0)
    model = O.CGN(ocfg, O.synthetic_word_embed(ocfg)).double()
    opts = O.make_optimizers(model)
    data = O.synthetic_batch(ocfg, 4, 5, seed=3)
    data = {k: (v.double() if v.is_floating_point() else v) for k, v in data.items()}
    cap = {}
    O.SELECT = select
    try:
        out = O.wake_sleep_step(model, opts, data, 5000, calc_z_loss=True, capture=cap)
    finally:
        O.SELECT = None
    return out, cap, model


def test_reporting_does_not_change_the_step():
    (d0, c0, e0), cap0, m0 = _run(None)
    sel = O.Selections()
    (d1, c1, e1), cap1, m1 = _run(sel)
    assert torch.equal(c0, c1)
    for a, b in zip(list(d0) + [e0[i] for i in (0, 1, 2, 3)], list(d1) + [e1[i] for i in (0, 1, 2, 3)]):
        assert abs(float(a) - float(b)) <= 1e-13 * max(1.0, abs(float(a)))
    for k in ("disc_grads", "gen_grads", "enc_grads"):
        for n, g in cap0[k].items():
            assert (g - cap1[k][n]).abs().max().item() <= 1e-12 * max(1.0, g.abs().max().item()), (k, n)
    for (n, p), (_, q) in zip(m0.named_parameters(), m1.named_parameters()):
        assert (p - q).abs().max().item() <= 1e-14, n
    # every kind of site was visited, in every scope of the step
    seen = set(sel.seen)
    for site in ("enc/hw0", "enc/hw1", "recon/hw1", "enc/tdnn2", "disc_l0/conv0", "disc_g0/conv2", "soft0/conv1",
                 "infer/conv0"):
        assert site in seen, (site, sorted(seen))
    assert not sel.adopted and not sel.refused


def test_only_tied_choices_are_adopted():
    sel = O.Selections(tie=1e-5)
    pre = torch.tensor([[1.0, -2.0, 1e-9, -1e-9]], dtype=torch.float64, requires_grad=True)
    forced = torch.tensor([[False, True, False, True]])  # flips all four gates; only the two tied ones may be taken
    sel.forced["s/r"] = forced
    sel.scope = "s"
    y = sel.relu("r", pre)
    y.sum().backward()
    assert pre.grad.tolist() == [[1.0, 0.0, 0.0, 1.0]]
    assert sel.adopted == {"s/r": 2} and sel.refused == {"s/r": 2}

    sel = O.Selections(tie=1e-5)
    v = torch.tensor([[[0.5, 0.5 - 1e-9, 0.1], [0.3, 0.9, 0.2], [-0.2, -1e-10, -0.5]]], dtype=torch.float64,
                     requires_grad=True)  # [N=1, C=3, P=3]
    sel.forced["s/m"] = torch.tensor([[1, 0, 1]])  # map 0: tied winner (taken); map 1: not tied (refused); map 2: gate
    sel.scope = "s"
    out = sel.maxpool("m", v, relu=True)
    out.sum().backward()
    assert v.grad[0, 0].tolist() == [0.0, 1.0, 0.0]
    assert v.grad[0, 1].tolist() == [0.0, 1.0, 0.0]
    assert v.grad[0, 2].tolist() == [0.0, 1.0, 0.0]  # max is -1e-10: the oracle says closed, the fp32 pass open (tied)
    assert sel.adopted == {"s/m:winner": 1, "s/m:gate": 1} and sel.refused == {"s/m:winner": 1}
