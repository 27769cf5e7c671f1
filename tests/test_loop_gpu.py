"""GPU: the sleep-phase sampler, the whole training loop and checkpoints: the batched on-device beam sampler (hip/sampler.py, cgn_beam_topk_f32) against the oracle's restatement of
sample_from_generator / Beam (model/cgn.py:747-921, utils/beam_search.py) in fp64 on identical weights and seeds:
identical token sequences, joint log-probabilities within 1e-4."""
import pytest
import torch

from oracle import cgn_oracle as O

pytestmark = pytest.mark.gpu

CFG = dict(max_word_len=12, max_seq_len=12, word_vocab_size=1200, char_vocab_size=64, encoder_rnn_size=64,
           latent_variable_size=32, decoder_rnn_size=96, sentiment_kernel_num=20)


def _build(dev, eos_bias=0.0, seed=5, scale=3.0):
    import ldrs_amd  # noqa: F401
    from ldrs_amd.model.cgn import Controlled_Generation_Sentence
    from ldrs_amd.utils.config import Config
    from ldrs_amd.utils.synthetic import SyntheticDataHandler
    kw = dict(CFG)
    pos = [kw.pop(k) for k in ("max_word_len", "max_seq_len", "word_vocab_size", "char_vocab_size")]
    cfg = Config(*pos, 5e-5, 7.9, 0.1, 0.1, 1.0, **kw)
    ocfg = O.OracleConfig(**CFG)
    torch.manual_seed(seed)
    emb = O.synthetic_word_embed(ocfg)
    oracle = O.CGN(ocfg, emb).double()
    with torch.no_grad():
        oracle.generator.fc.weight.mul_(scale)  # a peaked vocabulary distribution: well separated candidates
        oracle.generator.fc.bias[ocfg.word_vocab_size - 1] += eos_bias
    model = Controlled_Generation_Sentence(cfg, emb)
    model.load_state_dict({k: v.float() for k, v in oracle.state_dict().items()})
    model = model.to(dev)
    return cfg, ocfg, oracle, model, SyntheticDataHandler(cfg, 8, 10)


@pytest.mark.parametrize("graphed", [True, False])
@pytest.mark.parametrize("S,K,eos_bias", [(6, 5, 0.0), (10, 10, 2.5), (3, 1, 0.0), (7, 4, 4.0)])
def test_beam_sampler_matches_oracle(dev, S, K, eos_bias, graphed, monkeypatch):
    from ldrs_amd.hip import sampler as hs
    orig_init = hs.BeamSampler.__init__
    monkeypatch.setattr(hs.BeamSampler, "__init__", lambda self, eng, use_graph=True: orig_init(self, eng, graphed))
    cfg, ocfg, oracle, model, handler = _build(dev, eos_bias)
    V = ocfg.word_vocab_size
    pad, go, eos = V - 3, V - 2, V - 1
    g = torch.Generator().manual_seed(100 + S)
    seed = torch.cat([torch.randn(S, ocfg.latent_variable_size, generator=g),
                      torch.bernoulli(torch.full((S, 1), 0.5), generator=g)], 1)
    ref_h, ref_s = O.sample_from_generator(oracle, seed.double(), ocfg.max_seq_len, K, go, eos, pad, n_best=min(2, K))
    codes, scores = model.sample_from_generator(handler.gen_batch_loader, cfg.max_seq_len, seed, True, beam_size=K,
                                                n_best=min(2, K), samples=S)
    assert len(codes) == S
    finished = 0
    for s in range(S):
        got = [list(h) for h in zip(*codes[s])]  # back from the zip(*hyps) layout of cgn.py:879
        assert got == ref_h[s], (s, got, ref_h[s])
        finished += ref_h[s][0][-1] == eos
        for a, b in zip(scores[s].tolist(), ref_s[s].tolist()):
            assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (s, a, b)
    if eos_bias >= 2.5:
        assert 0 < finished, "no sentence finished early: the frozen-beam path was not exercised"


def test_sample_generator_for_learning_api(dev):
    """cgn.py:993-1034 / train_functions.py:7-17: lists of word ids + the codes c, usable as a discriminator batch."""
    import ldrs_amd  # noqa: F401
    from ldrs_amd import train_functions
    cfg, ocfg, oracle, model, handler = _build(dev)
    sents, c = train_functions.generate_samples_for_disc_training(model, handler, cfg, True, 8, 4)
    assert len(sents) == 8 and tuple(c.shape) == (8, 1)
    assert all(1 <= len(s) <= cfg.max_seq_len and all(0 <= w < ocfg.word_vocab_size for w in s) for s in sents)
    batch = handler.create_sentiment_batch(sents)
    assert batch.shape[0] == 8


def test_train_complete_model_runs_the_whole_wake_sleep_loop(dev):
    """train_functions.py:19-108 end to end: sampler -> discriminator step -> c inference -> encoder/generator step."""
    import ldrs_amd  # noqa: F401
    from ldrs_amd import train_functions
    from ldrs_amd.utils.synthetic import SyntheticDataHandler
    cfg, ocfg, oracle, model, _ = _build(dev)
    handler = SyntheticDataHandler(cfg, 20, cfg.max_seq_len - 1, num_batches=3)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    d, eg = train_functions.train_complete_model(model, handler, cfg, True, 0.3, None, None, "t", num_steps=3)
    vals = [float(x) for x in d] + [float(x) for x in eg]
    assert all(v == v and abs(v) < 1e6 for v in vals), vals
    moved = [n for n, p in model.named_parameters() if not torch.equal(p.detach(), before[n])]
    assert any(n.startswith("generator.") for n in moved) and any(n.startswith("sentiment_discriminator.") for n in moved)
    assert "embedding.word_embed.weight" not in moved


def test_wake_sleep_loop_on_a_text_corpus(dev):
    """The whole loop on real text through the in-memory BatchLoader / DataHandler (ragged sentences: reversed,
    left-padded encoder batches, padded discriminator batches, beam-sampled generated batch)."""
    import numpy as np
    import ldrs_amd  # noqa: F401
    from ldrs_amd import train_functions
    from ldrs_amd.model.cgn import Controlled_Generation_Sentence
    from ldrs_amd.utils.config import Config
    from ldrs_amd.utils.data_handler import DataHandler
    rng = np.random.RandomState(0)
    words = ["w%d" % i for i in range(60)]
    sents = [" ".join(rng.choice(words, size=rng.randint(6, 12))) for _ in range(48)]
    labelled = [" ".join(rng.choice(words, size=rng.randint(6, 12))) for _ in range(60)]
    handler = DataHandler(sents, labelled, rng.randint(0, 2, size=60).tolist(), batch_size=10, train_lines=40)
    bl = handler.gen_batch_loader
    cfg = Config(bl.max_word_len, bl.max_seq_len, bl.words_vocab_size, bl.chars_vocab_size, 1e-3, 7.9, 0.1, 0.1, 1.0,
                 word_embed_size=24, kernels=((1, 4), (2, 6)), encoder_rnn_size=32, latent_variable_size=8,
                 decoder_rnn_size=32, sentiment_kernel_size=(3, 4, 5), sentiment_kernel_num=6)
    torch.manual_seed(1)
    emb = (torch.rand(bl.words_vocab_size, 24) * 0.5 - 0.25).numpy()
    model = Controlled_Generation_Sentence(cfg, emb).to(dev)
    d, eg = train_functions.train_complete_model(model, handler, cfg, True, 0.3, None, None, "t", num_steps=4)
    vals = [float(x) for x in d] + [float(x) for x in eg]
    assert all(v == v and abs(v) < 1e6 for v in vals), vals


def test_checkpoint_round_trip_through_state_dict(dev):
    """t.save(cgn_model.state_dict()) / load_state_dict (train_functions.py:107-108, main.py:216): the parameters are
    views into the flat arenas (padded row strides included), the checkpoint must still carry the reference's keys and
    shapes, reload into a fresh model (before or after its step engine exists) and continue training identically."""
    import io
    cfg, ocfg, oracle, model, handler = _build(dev)
    eng = model.step_engine()
    opts = dict(disc=eng.new_optimizer("disc"), enc=eng.new_optimizer("enc"), gen=eng.new_optimizer("gen"))
    data = {k: v.to(dev).contiguous() for k, v in O.synthetic_batch(ocfg, 6, 9, seed=3).items()}
    eng.wake_sleep(opts, data, 0.5, 0.9)
    buf = io.BytesIO()
    torch.save(model.state_dict(), buf)
    ref_keys = set(oracle.state_dict().keys())
    sd = torch.load(io.BytesIO(buf.getvalue()), weights_only=True)
    assert set(sd.keys()) == ref_keys
    assert all(tuple(sd[k].shape) == tuple(v.shape) for k, v in oracle.state_dict().items())
    losses = []
    for late_engine in (False, True):
        _, _, _, m2, _ = _build(dev, seed=99)  # different initial weights
        if late_engine:
            m2.step_engine()  # arenas already built: load_state_dict must write through the views
        m2.load_state_dict(sd)
        for (n, p), (_, q) in zip(model.named_parameters(), m2.named_parameters()):
            assert torch.equal(p, q), n
        e2 = m2.step_engine()
        o2 = dict(disc=e2.new_optimizer("disc"), enc=e2.new_optimizer("enc"), gen=e2.new_optimizer("gen"))
        dl, c, eg = e2.wake_sleep(o2, data, 0.5, 0.9)
        losses.append([float(x) for x in dl] + [float(x) for x in eg[:4]])
    assert losses[0] == losses[1]


def test_timed_out_persistent_launch_stops_the_training_loop(dev, tmp_path):
    """A persistent LSTM launch whose bounded wait ran out leaves a sticky device word (include/cgn_hip.h); the product
    path must not train on, log or checkpoint what came after it. The word is forced here (a healthy GPU never sets it)
    and every host sync point of the loop is expected to raise: the logging points, the checkpoint, the end of the
    loop, the sampler."""
    import ldrs_amd  # noqa: F401
    from ldrs_amd import train_functions
    from ldrs_amd.hip import ops
    from ldrs_amd.utils.synthetic import SyntheticDataHandler
    cfg, ocfg, oracle, model, _ = _build(dev)
    handler = SyntheticDataHandler(cfg, 20, cfg.max_seq_len - 1, num_batches=3)
    train_functions.train_complete_model(model, handler, cfg, True, 0.3, None, None, "t", num_steps=1)  # creates the scratch
    assert ops.lstm_sticky_timeouts() == 0 and len(ops._lstm_sync) > 0
    try:
        for ws in ops._lstm_sync.values():
            ws[-32] = 1  # what a timed-out launch writes (persist_sync_bytes(): the last 128-byte line)
        assert ops.lstm_sticky_timeouts() > 0
        with pytest.raises(ops.LstmTimeoutError):
            ops.raise_on_lstm_timeout()
        with pytest.raises(ops.LstmTimeoutError):  # quiet loop: the check at the end of the loop
            train_functions.train_complete_model(model, handler, cfg, True, 0.3, None, None, "t", num_steps=2)
        with pytest.raises(ops.LstmTimeoutError):  # logging loop: the first emit()
            train_functions.train_complete_model(model, handler, cfg, True, 0.3, None, None, "t", num_steps=2,
                                                 log_every=1, validate=False)
        with pytest.raises(ops.LstmTimeoutError):  # the sampler's own sync point
            train_functions.generate_samples_for_disc_training(model, handler, cfg, True, 4, 4)
        train_functions.CHECKPOINT_EVERY, keep = 1, train_functions.CHECKPOINT_EVERY
        try:
            with pytest.raises(ops.LstmTimeoutError):  # never write weights trained on a timed-out launch
                train_functions.train_complete_model(model, handler, cfg, True, 0.3, None, str(tmp_path), "t", num_steps=1)
            assert not list(tmp_path.iterdir())
        finally:
            train_functions.CHECKPOINT_EVERY = keep
    finally:
        ops.clear_lstm_sticky_timeouts()
    assert ops.lstm_sticky_timeouts() == 0
    # the launches that ran while the word was set gave up at their first wait, and still left their counters at zero
    assert ops.lstm_counters_at_rest(dev)
    train_functions.train_complete_model(model, handler, cfg, True, 0.3, None, None, "t", num_steps=1)


class _Writer:
    def __init__(self):
        self.rows = []

    def add_scalar(self, tag, value, step):
        self.rows.append((tag, float(value), step))


def test_validation_loops_follow_the_reference_schedule(dev, monkeypatch):
    """train_functions.py:152-263: the two validation loops, run by train_complete_model on the reference's schedule
    (discriminator every 40th, encoder/generator every 100th iteration, both at iteration 0) and stand-alone; forward
    only: no parameter moves, the model is back in training mode afterwards, the logged names are the reference's."""
    import ldrs_amd  # noqa: F401
    from ldrs_amd import train_functions
    from ldrs_amd.utils.synthetic import SyntheticDataHandler
    cfg, ocfg, oracle, model, _ = _build(dev)
    handler = SyntheticDataHandler(cfg, 10, cfg.max_seq_len - 1, num_batches=3, val_lines=20, num_dev_batches=2)
    monkeypatch.setattr(train_functions, "ENC_GEN_VALID_BATCH", 10)  # the reference's 500 (:165) needs 500 validation lines
    model.discriminator_mode(train_mode=True), model.encoder_mode(train_mode=True), model.generator_mode(train_mode=True)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    w = _Writer()
    out = train_functions.enc_gen_validation(model, handler, True, 7, model.valid_encoder_generator(), w, batch_size=10,
                                             verbose=False)
    assert set(out) == {"encoder_loss", "generator_loss", "generator_z_loss", "generator_c_loss", "cross_entropy", "kld",
                        "kld_coef", "temp_coef"}
    assert all(v == v for v in out.values()) and out["cross_entropy"] > 0 and out["kld"] >= 0
    out_d = train_functions.disc_validation(model, handler, cfg, True, 7, handler.num_dev_sentiment_batches,
                                            model.discriminator_sentiment_valid(handler, True, return_accuracy=True), 10,
                                            w, verbose=False)
    assert set(out_d) == {"labeled_cross_entropy", "labeled_accuracy", "generated_cross_entropy", "generated_accuracy"}
    assert 0.0 <= out_d["labeled_accuracy"] <= 100.0 and 0.0 <= out_d["generated_accuracy"] <= 100.0
    assert all(torch.equal(p.detach(), before[n]) for n, p in model.named_parameters()), "validation moved a parameter"
    assert {t.split("/")[0] for t, _, _ in w.rows} == {"valid_enc_gen", "valid_discriminator"}
    # inside the loop: iteration 0 runs both validations, iteration 1 neither; training still moves the weights
    w2 = _Writer()
    train_functions.train_complete_model(model, handler, cfg, True, 0.3, w2, None, "t", num_steps=2)
    tags = [(t, s) for t, _, s in w2.rows]
    assert ("valid_discriminator/labeled_accuracy", 0) in tags and ("valid_enc_gen/cross_entropy", 0) in tags
    assert not any(t.startswith("valid_") and s == 1 for t, s in tags)
    assert ("train_discriminator/total_loss", 0) in tags and ("train_enc_gen/encoder_loss", 0) in tags
    assert any(not torch.equal(p.detach(), before[n]) for n, p in model.named_parameters())


def _greedy_margins(oracle, zc, tokens, go):
    """fp64 oracle, teacher-forced along `tokens`: log-probability gap between the best and the second-best word at
    every decoding step (generator.py:30-41, cgn.py:793-803 with beam size 1)."""
    import torch.nn.functional as F
    gen, emb = oracle.generator, oracle.embedding.word_embed
    margins, state, prev = [], None, go
    with torch.no_grad():
        for tok in tokens:
            x = torch.cat([emb(torch.tensor([prev])).unsqueeze(1), zc.view(1, 1, -1)], 2)
            out, state = gen.rnn(x, state)
            top = F.log_softmax(gen.fc(out.squeeze(1)), dim=1)[0].topk(2).values
            margins.append(float(top[0] - top[1]))
            prev = tok
    return margins


def test_bf16_mode_decodes_identical_greedy_tokens(dev):
    """North-star's optional bf16 mode (MFMA operands rounded to bf16, fp32 accumulation and storage): greedy decoding
    (beam size 1) of the same codes on the same weights picks the SAME tokens as the fp32 mode wherever the choice is
    not a near-tie. Near-tie = the fp64 oracle, teacher-forced along the fp32 decode, sees the two best words closer
    than 5e-3 nats (bf16 operands move a logit by ~3e-3 of its size - these random-weight logits are O(0.3) - and the
    recurrent state by as much per step); behind the
    first near-tie of a sentence the two decodes may legitimately walk different paths. Random weights tie often, so
    the test also demands that most sentences have a clear prefix of at least 3 tokens (it is not vacuous)."""
    from ldrs_amd.hip import ops
    S, TIE = 16, 5e-3
    out = {}
    for mode in ("f32", "bf16"):
        prev = ops.set_compute(mode)
        try:
            cfg, ocfg, oracle, model, handler = _build(dev)  # a fresh model per mode: decode graphs bake the mode in
            seed = torch.cat([torch.randn(S, ocfg.latent_variable_size, generator=torch.Generator().manual_seed(77)),
                              torch.bernoulli(torch.full((S, 1), 0.5), generator=torch.Generator().manual_seed(78))], 1)
            codes, scores = model.sample_from_generator(handler.gen_batch_loader, cfg.max_seq_len, seed, True, beam_size=1,
                                                        n_best=1, samples=S)
            out[mode] = [[int(h[0]) for h in codes[s]] for s in range(S)]
        finally:
            ops.set_compute(prev)
    go = ocfg.word_vocab_size - 2
    clear_prefix, compared, identical = [], 0, 0
    for s in range(S):
        ref, got = out["f32"][s], out["bf16"][s]
        m = _greedy_margins(oracle, seed[s].double(), ref, go)
        n_clear = next((i for i, v in enumerate(m) if v < TIE), len(ref))
        clear_prefix.append(n_clear)
        assert got[:n_clear] == ref[:n_clear], (s, n_clear, ref, got, m)
        compared += n_clear
        identical += got == ref
    assert sum(n >= 3 for n in clear_prefix) >= S // 2, clear_prefix
    print("bf16 greedy decode: %d of %d sentences identical to fp32; %d tokens compared before a near-tie" %
          (identical, S, compared))


@pytest.mark.parametrize("graphed", [True, False])
@pytest.mark.parametrize("S,K,eos_bias", [(6, 5, 0.0), (7, 4, 4.0), (3, 1, 0.0)])
def test_beam_sampler_learning_returns_the_parent_beams_log_softmax_rows(dev, S, K, eos_bias, graphed, monkeypatch):
    """learning=True (cgn.py:880,901-902; Beam.get_hyp_probs, beam_search.py:140-152): per sentence the [len, V]
    log_softmax rows of the parent beams along beam 0's back-pointer chain, against the oracle's restatement in fp64
    (absolute 1e-4 on log-probabilities; same lengths, i.e. the same early finishes)."""
    from ldrs_amd.hip import sampler as hs
    orig_init = hs.BeamSampler.__init__
    monkeypatch.setattr(hs.BeamSampler, "__init__", lambda self, eng, use_graph=True: orig_init(self, eng, graphed))
    cfg, ocfg, oracle, model, handler = _build(dev, eos_bias)
    V = ocfg.word_vocab_size
    pad, go, eos = V - 3, V - 2, V - 1
    g = torch.Generator().manual_seed(100 + S)
    seed = torch.cat([torch.randn(S, ocfg.latent_variable_size, generator=g),
                      torch.bernoulli(torch.full((S, 1), 0.5), generator=g)], 1)
    ref = O.sample_from_generator(oracle, seed.double(), ocfg.max_seq_len, K, go, eos, pad, learning=True)
    got = model.sample_from_generator(handler.gen_batch_loader, cfg.max_seq_len, seed, True, beam_size=K, samples=S,
                                      learning=True)
    assert len(got) == S
    for s in range(S):
        assert tuple(got[s].shape) == tuple(ref[s].shape) == (ref[s].shape[0], V), (s, got[s].shape, ref[s].shape)
        err = (got[s].double().cpu() - ref[s]).abs().max().item()
        assert err <= 1e-4, (s, err)
        # each row is a normalised distribution
        assert (got[s].double().exp().sum(1) - 1).abs().max().item() <= 1e-4
    # a second call on the same (S, K) state reuses the captured graphs and gives the same rows
    again = model.sample_from_generator(handler.gen_batch_loader, cfg.max_seq_len, seed, True, beam_size=K, samples=S,
                                        learning=True)
    for a, b in zip(got, again):
        assert torch.equal(a, b)
