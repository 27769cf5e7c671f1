"""Generates tests/golden/reference_modules.npz by RUNNING the reference's own modules (build container only).

Run as:  python -O tests/golden/make_golden.py       (-O: model/encoder.py:35 asserts on a helper that is not
torch-2.x clean, utils/functional.py:13-15)

What runs from /root/reference (imported, never copied): model/generator.py Generator.forward,
selfModules/highway.py Highway.forward, model/encoder.py Encoder.forward, utils/functional.py kld_coef/temp_coef,
utils/config.py Config, and model/discriminator_sentiment.py Sentiment_CNN.forward — that file is Python-2 syntax
(print statements), so it is converted IN MEMORY with lib2to3 and exec'd; nothing is written to disk.
selfModules/tdnn.py TDNN.forward does not run under torch 2.x (keepdim-era squeeze, tdnn.py:43) and cgn.py's step
closures are run by tests/golden/make_golden_step.py (two textual substitutions, listed there).

The .npz holds only data: seeds-derived inputs, the modules' state_dicts and their outputs.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_modules.npz")


def load_sentiment_cnn():
    from lib2to3 import refactor
    src = open(os.path.join(REF, "model", "discriminator_sentiment.py")).read()
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package("lib2to3.fixes"))
    py3 = str(tool.refactor_string(src + "\n", "discriminator_sentiment.py"))
    mod = types.ModuleType("ref_discriminator_sentiment")
    exec(compile(py3, "discriminator_sentiment(2to3)", "exec"), mod.__dict__)
    return mod.Sentiment_CNN


def main():
    if __debug__:
        raise SystemExit("run with python -O (see module docstring)")
    sys.path.insert(0, REF)
    from model.encoder import Encoder
    from model.generator import Generator
    from selfModules.highway import Highway
    from utils.functional import kld_coef, temp_coef
    from utils.config import Config
    import torch.nn.functional as F
    import contextlib
    import io

    out = {}
    cfg = Config(max_word_len=7, max_seq_len=5, word_vocab_size=50, char_vocab_size=20, learning_rate=5e-5,
                 lambda_c=7.9, lambda_z=0.1, lambda_u=0.1, beta=1.0)
    out["config_sum_depth"] = np.array(cfg.sum_depth)
    out["config_max_seq_len"] = np.array(cfg.max_seq_len)
    # shrink the hard-coded dims (config.py:14-31) so the fixture stays small
    cfg.word_embed_size, cfg.encoder_rnn_size, cfg.latent_variable_size = 12, 16, 6
    cfg.decoder_rnn_size, cfg.decoder_num_layers = 20, 2
    cfg.kernels = [(1, 3), (2, 5), (3, 4)]
    cfg.sum_depth = 12
    cfg.sentiment_kernel_size, cfg.sentiment_kernel_num = [3, 4, 5], 6
    out["dims"] = np.array([cfg.word_embed_size, cfg.sum_depth, cfg.encoder_rnn_size, cfg.latent_variable_size,
                            cfg.decoder_rnn_size, cfg.word_vocab_size, cfg.sentiment_kernel_num])

    torch.manual_seed(20171110)
    B, L, T = 3, 5, 6

    # --- Highway (selfModules/highway.py)
    hw = Highway(24, 2, F.relu)
    x = torch.randn(B * L, 24)
    for k, v in hw.state_dict().items():
        out["highway/" + k] = v.numpy()
    out["highway_in"] = x.numpy()
    out["highway_out"] = hw(x).detach().numpy()

    # --- Encoder (model/encoder.py)
    enc = Encoder(cfg)
    xin = torch.randn(B, L, cfg.word_embed_size + cfg.sum_depth)
    ctx, h_n, c_n = enc(xin, None)
    for k, v in enc.state_dict().items():
        out["encoder/" + k] = v.numpy()
    out["encoder_in"] = xin.numpy()
    out["encoder_ctx"], out["encoder_h_n"], out["encoder_c_n"] = (t.detach().numpy() for t in (ctx, h_n, c_n))

    # --- Generator (model/generator.py); drop_prob = 0 so F.dropout is the identity under every torch version
    gen = Generator(cfg)
    gin = torch.randn(B, T, cfg.word_embed_size)
    zc = torch.randn(B, cfg.latent_variable_size + 1)
    logits, (gh, gc) = gen(gin, zc, 0.0, None)
    for k, v in gen.state_dict().items():
        out["generator/" + k] = v.numpy()
    out["generator_in"], out["generator_zc"] = gin.numpy(), zc.numpy()
    out["generator_logits"], out["generator_h_n"], out["generator_c_n"] = (t.detach().numpy()
                                                                         for t in (logits, gh, gc))

    # --- Sentiment_CNN (model/discriminator_sentiment.py, via in-memory lib2to3)
    Sentiment_CNN = load_sentiment_cnn()
    with contextlib.redirect_stdout(io.StringIO()):
        cnn = Sentiment_CNN(cfg)
    cnn.eval()
    sx = torch.randn(B, L, cfg.word_embed_size)
    for k, v in cnn.state_dict().items():
        out["cnn/" + k] = v.numpy()
    out["cnn_in"] = sx.numpy()
    out["cnn_logit"] = cnn(sx).detach().numpy()
    try:
        cnn(torch.randn(B, 4, cfg.word_embed_size))
        out["cnn_short_input_raises"] = np.array(0)
    except RuntimeError:
        out["cnn_short_input_raises"] = np.array(1)

    # --- schedules (utils/functional.py), evaluated under Python 3 semantics (true division)
    steps = np.array([0, 1, 999, 3499, 3500, 3501, 5000, 10000, 200000, 2500000])
    out["sched_steps"] = steps
    out["sched_kld_py3"] = np.array([kld_coef(int(s)) for s in steps])
    out["sched_kld_ext_py3"] = np.array([kld_coef(int(s), True) for s in steps])
    out["sched_temp"] = np.array([temp_coef(int(s)) for s in steps])

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "with", len(out), "arrays")


if __name__ == "__main__":
    main()
