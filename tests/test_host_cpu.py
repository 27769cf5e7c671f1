"""CPU checks of the host side that mirrors the reference's module API (no kernel runs here):
state_dict keys / parameter shapes against the golden fixture taken from the reference's own modules, schedules,
Config, the synthetic batch layout, the host-only GEMM planner, and the "no GPU => loud failure" rule."""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch

import ldrs_amd  # noqa: F401
from ldrs_amd.hip import _lib
from ldrs_amd.model.cgn import Controlled_Generation_Sentence
from ldrs_amd.model.discriminator_sentiment import Sentiment_CNN
from ldrs_amd.model.encoder import Encoder
from ldrs_amd.model.generator import Generator
from ldrs_amd.selfModules.highway import Highway
from ldrs_amd.utils import functional as Fn
from ldrs_amd.utils import synthetic
from ldrs_amd.utils.config import Config

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_modules.npz"))


def golden_config():
    E, S, He, Z, Hg, V, Co = (int(v) for v in G["dims"])
    return Config(7, 5, V, 20, 5e-5, 7.9, 0.1, 0.1, 1.0, word_embed_size=E, kernels=((1, 3), (2, 5), (3, 4)),
                  encoder_rnn_size=He, latent_variable_size=Z, decoder_rnn_size=Hg, sentiment_kernel_num=Co)


def _golden_shapes(prefix):
    return {k[len(prefix):]: tuple(G[k].shape) for k in G.files if k.startswith(prefix)}


@pytest.mark.parametrize("prefix,make", [
    ("encoder/", lambda c: Encoder(c)),
    ("generator/", lambda c: Generator(c)),
    ("cnn/", lambda c: Sentiment_CNN(c)),
    ("highway/", lambda c: Highway(24, 2, torch.relu)),
])
def test_state_dict_keys_and_shapes_match_reference_modules(prefix, make):
    """The fixture holds the state_dicts of the reference's own Encoder/Generator/Sentiment_CNN/Highway."""
    mod = make(golden_config())
    ours = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
    assert ours == _golden_shapes(prefix)
    # and the reference's weights load straight in (checkpoint compatibility)
    mod.load_state_dict({k[len(prefix):]: torch.from_numpy(G[k]) for k in G.files if k.startswith(prefix)}, strict=True)


def test_full_model_state_dict_layout(tmp_path):
    cfg = golden_config()
    path = tmp_path / "word_embeddings.npy"
    np.save(path, synthetic.word_embeddings(cfg))
    model = Controlled_Generation_Sentence(cfg, str(path))
    keys = set(model.state_dict())
    expect = {"embedding.word_embed.weight", "embedding.char_embed.weight", "e2mu.weight", "e2mu.bias",
              "e2logvar.weight", "e2logvar.bias"}
    expect |= {"embedding.TDNN.TDNN_kernel-%d" % i for i in range(3)}
    expect |= {"encoder." + k for k in _golden_shapes("encoder/")}
    expect |= {"generator." + k for k in _golden_shapes("generator/")}
    expect |= {"sentiment_discriminator." + k for k in _golden_shapes("cnn/")}
    assert keys == expect
    assert not model.embedding.word_embed.weight.requires_grad  # embedding.py:23
    # optimizer groups (cgn.py:676-686, :1036-1037)
    n_enc = len(list(model.encoder_params()))
    assert n_enc == len([k for k in keys if not k.startswith(("generator.", "sentiment_discriminator."))]) - 1
    assert len(list(model.generator_params())) == len(_golden_shapes("generator/"))


def test_step_closures_fail_loudly_without_gpu(tmp_path):
    """No CPU fallback: on a host without an MI355X the fused step raises instead of computing something else."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = golden_config()
    path = tmp_path / "word_embeddings.npy"
    np.save(path, synthetic.word_embeddings(cfg))
    model = Controlled_Generation_Sentence(cfg, str(path))
    with pytest.raises(RuntimeError, match="MI355X"):
        model.step_engine()


def test_schedules_match_reference_outputs():
    steps = G["sched_steps"]
    np.testing.assert_allclose([Fn.kld_coef(int(s), py2_floor=False) for s in steps], G["sched_kld_py3"], rtol=1e-12)
    np.testing.assert_allclose([Fn.kld_coef(int(s), True, py2_floor=False) for s in steps], G["sched_kld_ext_py3"],
                               rtol=1e-12)
    np.testing.assert_allclose([Fn.temp_coef(int(s)) for s in steps], G["sched_temp"], rtol=1e-12)
    assert Fn.kld_coef(3500) == 0.5 and Fn.temp_coef(0) == 1.0 and Fn.temp_coef(200000) == 0.1
    assert math.isclose(Fn.kld_coef(0), (math.tanh(-4) + 1) / 2)  # Python-2 floor division, functional.py:31
    assert math.isclose(Fn.kld_coef(5000), (math.tanh(1) + 1) / 2)
    assert math.isclose(Fn.kld_coef(5000, py2_floor=False), (math.tanh(1.5) + 1) / 2)


def test_config_known_answers_and_validation():
    c = Config(12, 20, 20000, 64, 5e-5, 7.9, 0.1, 0.1, 1.0)
    assert c.sum_depth == 525 == int(G["config_sum_depth"])
    assert c.max_seq_len == 21  # config.py:9
    assert (c.encoder_rnn_size, c.decoder_rnn_size, c.latent_variable_size) == (600, 800, 1100)  # reference defaults
    with pytest.raises(ValueError):
        Config(5, 20, 100, 10, 1e-3, 1, 1, 1, 1)  # word shorter than the widest TDNN kernel
    with pytest.raises(ValueError):
        Config(12, 20, 100, 10, 1e-3, 1, 1, 1, 1, encoder_rnn_size=30)


def test_synthetic_batch_layout():
    """utils/batch_loader.py:271-316: enc reversed, dec = [go]+s, target = s+[end], specials = last vocab indices."""
    cfg = golden_config()
    h = synthetic.SyntheticDataHandler(cfg, batch_size=4, seq_len=5)
    enc_w, enc_c, dec_w, dec_c, tgt = h.gen_batch_loader.next_batch(4, "train", 0)
    V = cfg.word_vocab_size
    assert enc_w.shape == (4, 5) and enc_c.shape == (4, 5, 7) and dec_w.shape == (4, 6) and tgt.shape == (4, 6)
    assert (dec_w[:, 0] == V - 2).all() and (tgt[:, -1] == V - 1).all()
    assert (dec_w[:, 1:] == tgt[:, :-1]).all() and (enc_w[:, ::-1] == tgt[:, :-1]).all()
    assert enc_w.max() < V - 3
    padded = h.create_sentiment_batch([[1, 2, 3], [4]])
    assert padded.tolist() == [[1, 2, 3], [4, V - 3, V - 3]]


def _plan(M, N, K, nt=True, **force):
    lib = _lib.load()
    d = _lib.GemmDesc()
    d.M, d.N, d.K = M, N, K
    d.a_m, d.a_k = _lib.Axis(K, 0, 0), _lib.Axis(1, 0, 0)
    d.b_k, d.b_n = (_lib.Axis(1, 0, 0), _lib.Axis(K, 0, 0)) if nt else (_lib.Axis(N, 0, 0), _lib.Axis(1, 0, 0))
    d.ldc = N
    d.workspace, d.workspace_bytes = 0x1000, 96 << 20  # the planner only tests the pointer for NULL
    for k, v in force.items():
        setattr(d, k, v)
    out = [C.c_int32() for _ in range(4)]
    rc = lib.cgn_gemm_plan(C.byref(d), *[C.byref(o) for o in out])
    return rc, tuple(o.value for o in out)


def test_gemm_planner_host_only():
    """cgn_gemm_plan is pure host code: tile, split-K and operand contiguity for the cfg2 shapes."""
    rc, (tile, sk, ak, bk) = _plan(1344, 20000, 512)  # vocabulary projection
    assert rc == 0 and tile == 128 and sk == 1 and (ak, bk) == (1, 1)
    rc, (tile, sk, ak, bk) = _plan(1344, 512, 20000, nt=False)  # dH = dlogits . W_fc : long K, few tiles => split-K
    assert rc == 0 and sk > 1 and (ak, bk) == (1, 0)
    rc, (tile, sk, _, _) = _plan(64, 128, 1024)  # e2mu/e2logvar: 2 tiles, a latency chain => split hard
    assert rc == 0 and tile == 64 and sk == 8
    rc, (tile, sk, _, _) = _plan(1344, 2048, 512)  # LSTM input projection: 672 tiles already fill the GPU
    assert rc == 0 and tile == 64 and sk == 1
    rc, (tile, sk, _, _) = _plan(64, 64, 64)  # nothing to split
    assert rc == 0 and sk == 1
    rc, (_, sk, _, _) = _plan(1344, 512, 20000, nt=False, workspace=None)  # no workspace => no split, still valid
    assert rc == 0 and sk == 1
    rc, _ = _plan(1344, 512, 20000, nt=False, workspace=None, force_splitk=8)
    assert rc == _lib.CGN_ERR_WORKSPACE
    rc, _ = _plan(64, 64, 64, force_tile=96)
    assert rc == _lib.CGN_ERR_INVALID_ARG
    rc, (tile, sk, _, _) = _plan(256, 256, 4096, force_tile=64, force_splitk=4)
    assert rc == 0 and tile == 64 and sk == 4


def test_package_synthetic_inputs_equal_the_oracles():
    """bench.py's measured leg draws its inputs with ldrs_amd.utils.synthetic (no oracle import on that leg); the oracle's
    own generators (used by the parity tests) draw the same streams."""
    import numpy as np
    import torch
    import ldrs_amd  # noqa: F401
    from ldrs_amd.utils import synthetic
    from ldrs_amd.utils.config import Config
    from oracle import cgn_oracle as O
    for c_size in (1, 2):
        kw = dict(word_embed_size=12, kernels=((1, 3), (2, 5)), encoder_rnn_size=16, latent_variable_size=6,
                  decoder_rnn_size=20, sentiment_kernel_size=(3, 4), sentiment_kernel_num=5, c_size=c_size)
        cfg = Config(7, 5, 50, 20, 5e-5, 7.9, 0.1, 0.1, 1.0, **kw)
        ocfg = O.OracleConfig(max_word_len=7, max_seq_len=6, word_vocab_size=50, char_vocab_size=20, **kw)
        a, b = synthetic.synthetic_batch(cfg, 4, 5, seed=9), O.synthetic_batch(ocfg, 4, 5, seed=9)
        assert sorted(a) == sorted(b)
        for k in a:
            assert torch.equal(a[k], b[k]), k
        np.testing.assert_array_equal(synthetic.synthetic_word_embed(cfg), O.synthetic_word_embed(ocfg))
