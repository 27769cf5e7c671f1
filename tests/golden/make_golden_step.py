"""Generates tests/golden/reference_step.npz by RUNNING the reference's own step closures (build container only):
model/cgn.py  discriminator_sentiment_trainer().train (:269-341), discriminator_forward (:134-172) and
train_encoder_generator().train (:1049-1154), in the order train_functions.py:24-90 calls them, on the CPU, for two
consecutive wake-sleep iterations (the second one exercises Adam's moments and the updated weights); then the
validation closures valid_encoder_generator().valid (:1159-1250) and discriminator_sentiment_valid().valid (:174-240), and
on a fresh model the initial VAE pre-training closures initial_rvae_trainer / initial_rvae_valid (:105-132, :621-674).

Run as:  python -O tests/golden/make_golden_step.py      (-O: model/encoder.py:35 asserts on a helper that is not
torch-2.x clean, utils/functional.py:13-15)

How the Python-2 / old-torch sources are made to run under Python 3.10 / torch 2.10 — all IN MEMORY, nothing from the
reference is written to disk or into this repository:
  * model/cgn.py, model/discriminator_sentiment.py, selfModules/embedding.py hold Python-2 print statements: converted
    with lib2to3 and exec'd as modules of their own packages;
  * TWO textual substitutions, each the documented modern spelling of the same operation:
      - selfModules/tdnn.py:43   `x.max(2)[0].squeeze(2)` -> `x.max(2)[0]`   (torch < 0.2 kept the reduced dimension;
        the squeeze only removed it again; torch >= 0.2 reductions drop it themselves and the squeeze then raises);
      - model/cgn.py:1097,1147   `backward(retain_variables=True)` -> `backward(retain_graph=True)`  (argument renamed
        in torch 0.2, old name removed later);
  * for the beam sampler (sample_from_generator, cgn.py:747-921) FOUR more, none touching arithmetic: the two places
    where the reference hard-wires CUDA tensors (cgn.py:781 `cuda=True`, :845 `t.cuda.LongTensor`) use CPU tensors, and
    the two Python-2 integer divisions (model/generator.py:31, utils/beam_search.py:88) are written `//`;
  * model/encoder.py, selfModules/highway.py, utils/functional.py, utils/config.py are imported as they are.
Everything else is the reference's code executing: the optimizers it builds, its gradient routing (the z.grad alias,
generator step before the encoder backward), its losses. Schedules are evaluated under Python-3 division
(kld_coef: true division, i.e. the oracle's py2_floor=False). Generator dropout is 0.0 (F.dropout's default
`training` flag changed between torch versions; with p = 0 it is the identity under every version).

Random draws are recovered, not re-implemented: eps of encode_for_z (:96) by re-seeding torch's generator with the seed
set right before the call (nothing else draws in between: the text-CNN is in eval mode, every dropout has p = 0); the
text-CNN's training-mode dropout masks of the discriminator step with a forward hook on its nn.Dropout; the Bernoulli
draw of c (:170) is recorded as drawn, together with the probabilities it was drawn from.

The .npz holds only data: config dims, the initial state_dict, the batches, the recovered draws and the outputs
(losses, c, per-parameter gradients left in .grad by the closures, parameters after each iteration).
"""
import contextlib
import io
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
# name -> (file, Config positional args, dims that config.py hard-codes, batch size): "tiny" = 16-byte friendly sizes,
# "odd" = every row stride unaligned (E = 33), hidden sizes that are not multiples of 16, four TDNN widths
CASES = {
    "tiny": ("reference_step.npz", dict(max_word_len=7, max_seq_len=5, word_vocab_size=50, char_vocab_size=20),
             dict(word_embed_size=12, encoder_rnn_size=16, latent_variable_size=6, decoder_rnn_size=20,
                  kernels=[(1, 3), (2, 5), (3, 4)], sentiment_kernel_num=6), 4),
    "odd": ("reference_step_odd.npz", dict(max_word_len=8, max_seq_len=8, word_vocab_size=101, char_vocab_size=31),
            dict(word_embed_size=33, encoder_rnn_size=20, latent_variable_size=10, decoder_rnn_size=28,
                 kernels=[(1, 5), (2, 7), (3, 6), (6, 13)], sentiment_kernel_num=9), 7),
}


def exec_module(relpath, modname, package, subs=(), py2=True):
    src = open(os.path.join(REF, relpath)).read()
    for old, new in subs:
        assert src.count(old) >= 1, (relpath, old)
        src = src.replace(old, new)
    if py2:
        from lib2to3 import refactor
        tool = refactor.RefactoringTool(refactor.get_fixers_from_package("lib2to3.fixes"))
        src = str(tool.refactor_string(src + "\n", relpath))
    mod = types.ModuleType(modname)
    mod.__package__ = package
    mod.__file__ = relpath + " (in memory)"
    sys.modules[modname] = mod
    exec(compile(src, relpath + "(in memory)", "exec"), mod.__dict__)
    return mod


class Handler:
    """The two data_handler calls the discriminator closure makes (cgn.py:276, 282-283)."""

    def __init__(self):
        self.X = self.Y = None

    def get_sentiment_train_batch(self, i):
        return self.X, self.Y

    def create_sentiment_batch(self, sentences):
        return torch.tensor(sentences, dtype=torch.int64)


def main():
    if __debug__:
        raise SystemExit("run with python -O (see module docstring)")
    sys.path.insert(0, REF)
    import model  # noqa: F401  the reference's packages (their __init__ files are empty)
    import selfModules  # noqa: F401
    from utils.config import Config

    exec_module("selfModules/tdnn.py", "selfModules.tdnn", "selfModules",
                subs=[("x.max(2)[0].squeeze(2)", "x.max(2)[0]")], py2=False)
    # sampler only (the forward pass of Generator is untouched): Python-2 integer divisions
    exec_module("model/generator.py", "model.generator", "model",
                subs=[("(beam_batch_size/zc.size(0))", "(beam_batch_size//zc.size(0))")], py2=False)
    exec_module("utils/beam_search.py", "utils.beam_search", "utils",
                subs=[("prev_k = bestScoresId / num_words", "prev_k = bestScoresId // num_words")], py2=False)
    with contextlib.redirect_stdout(io.StringIO()):
        exec_module("selfModules/embedding.py", "selfModules.embedding", "selfModules")
        exec_module("model/discriminator_sentiment.py", "model.discriminator_sentiment", "model")
        cgn_mod = exec_module("model/cgn.py", "model.cgn", "model",
                              subs=[("backward(retain_variables=True)", "backward(retain_graph=True)"),
                                    # sampler: the reference hard-wires CUDA (cgn.py:781, :845); CPU tensors here
                                    ("Beam(beam_size, batch_loader, cuda=True)", "Beam(beam_size, batch_loader, cuda=False)"),
                                    ("active_idx = t.cuda.LongTensor(", "active_idx = t.LongTensor(")])

    for case in (sys.argv[1:] or list(CASES)):
        generate(cgn_mod, Config, *CASES[case])


def generate(cgn_mod, Config, fname, pos, dims, B):
    OUT = os.path.join(HERE, fname)
    cfg = Config(learning_rate=5e-5, lambda_c=7.9, lambda_z=0.1, lambda_u=0.1, beta=1.0, **pos)
    # shrink the hard-coded dims (config.py:14-31) so the fixture stays small
    for k, v in dims.items():
        setattr(cfg, k, v)
    cfg.decoder_num_layers = 2
    cfg.sum_depth = sum(od for _, od in cfg.kernels)
    cfg.sentiment_kernel_size = [3, 4, 5]
    out = {"dims": np.array([cfg.max_word_len, cfg.max_seq_len, cfg.word_vocab_size, cfg.char_vocab_size,
                             cfg.word_embed_size, cfg.encoder_rnn_size, cfg.latent_variable_size, cfg.decoder_rnn_size,
                             cfg.sentiment_kernel_num, cfg.char_embed_size]),
           "sentiment_dropout": np.array(cfg.sentiment_dropout), "learning_rate": np.array(cfg.learning_rate),
           "kernels": np.array(cfg.kernels)}

    g = torch.Generator().manual_seed(20171110)
    V, E = cfg.word_vocab_size, cfg.word_embed_size
    word_embed = ((torch.rand(V, E, generator=g) * 2 - 1) * 0.25).numpy()
    out["word_embed"] = word_embed
    torch.manual_seed(1)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "word_embeddings.npy")
        np.save(path, word_embed)
        with contextlib.redirect_stdout(io.StringIO()):
            cgn = cgn_mod.Controlled_Generation_Sentence(cfg, path)
    for k, v in cgn.state_dict().items():
        out["init/" + k] = v.detach().numpy().copy()

    handler = Handler()
    # train_functions.py:24-33
    cgn.discriminator_mode(train_mode=True)
    cgn.encoder_mode(train_mode=True)
    cgn.generator_mode(train_mode=True)
    disc_step = cgn.discriminator_sentiment_trainer(handler, False)
    enc_gen_step = cgn.train_encoder_generator(None)

    masks = []
    hook = cgn.sentiment_discriminator.dropout.register_forward_hook(
        lambda m, i, o: masks.append((i[0].detach().clone(), o.detach().clone())) if m.training else None)
    p_keep = 1.0 - cfg.sentiment_dropout

    L = cfg.max_seq_len - 1
    for it in range(2):
        tag = "it%d/" % it
        calc_z = it == 1  # iteration 0 as train_functions.py runs it (calc_z_loss=False), iteration 1 with the z pass
        s = torch.randint(0, V - 3, (B, L), generator=g)
        chars = torch.randint(0, cfg.char_vocab_size, (B, L, cfg.max_word_len), generator=g)
        go, end = V - 2, V - 1
        batch = [torch.flip(s, dims=[1]).contiguous().numpy(), torch.flip(chars, dims=[1]).contiguous().numpy(),
                 torch.cat([torch.full((B, 1), go, dtype=torch.int64), s], 1).numpy(),
                 np.zeros((B, L + 1, cfg.max_word_len), np.int64),  # decoder character input: converted, never used (:1058)
                 torch.cat([s, torch.full((B, 1), end, dtype=torch.int64)], 1).numpy()]
        x_lab = torch.randint(0, V - 3, (B, L), generator=g)
        y_lab = torch.randint(0, 2, (B,), generator=g)
        x_gen = torch.randint(0, V - 3, (B, L), generator=g)
        c_gen = torch.randint(0, 2, (B,), generator=g).float().view(B, 1)
        for nm, a in (("enc_word", batch[0]), ("enc_char", batch[1]), ("dec_word", batch[2]), ("target", batch[4]),
                      ("x_labelled", x_lab.numpy()), ("y_labelled", y_lab.numpy()), ("x_generated", x_gen.numpy()),
                      ("c_generated", c_gen.numpy())):
            out[tag + nm] = a
        gstep = 5000 + 1000 * it
        out[tag + "global_step"], out[tag + "calc_z"] = np.array(gstep), np.array(int(calc_z))

        # ---- discriminator step (train_functions.py:55-57)
        cgn.prep_disc_training()
        handler.X, handler.Y = x_lab.tolist(), y_lab.tolist()
        del masks[:]
        torch.manual_seed(100 + it)
        d_losses = disc_step(x_gen.tolist(), c_gen, 0)
        assert len(masks) == 2
        for nm, (i_, o_) in zip(("mask_labelled", "mask_generated"), masks):
            m = torch.where(i_ != 0, o_ / torch.where(i_ != 0, i_, torch.ones_like(i_)), torch.full_like(i_, 1.0 / p_keep))
            m = torch.where(m > 0.5 / p_keep, torch.full_like(m, 1.0 / p_keep), torch.zeros_like(m))  # exact {0, 1/keep}
            assert torch.allclose(i_ * m, o_, rtol=1e-6, atol=0)
            out[tag + nm] = m.numpy()
        out[tag + "disc_losses"] = np.array([float(x) for x in d_losses])
        for n, p in cgn.sentiment_discriminator.named_parameters():
            out[tag + "disc_grad/" + n] = p.grad.detach().numpy().copy()

        # ---- c inference (train_functions.py:81-82)
        cgn.discriminator_mode(train_mode=False)
        torch.manual_seed(200 + it)
        c_batch = cgn.discriminator_forward(batch, B, False)
        with torch.no_grad():  # the probabilities the draw was made from, by the reference's own modules (:163-167)
            p1 = torch.softmax(cgn.sentiment_discriminator(cgn.embedding.get_word_embed(torch.from_numpy(batch[0]))), 1)[:, 1]
        out[tag + "c"], out[tag + "c_p1"] = c_batch.numpy().copy(), p1.numpy().copy()

        # ---- encoder / generator step (train_functions.py:85-90)
        cgn.prep_enc_gen_training()
        seed = 300 + it
        torch.manual_seed(seed)
        eg = enc_gen_step(batch, B, False, 0.0, c_batch, gstep, calc_z_loss=calc_z)
        torch.manual_seed(seed)  # the same stream again: encode_for_z's randn calls (:96), in call order
        out[tag + "eps"] = torch.randn([B, cfg.latent_variable_size]).numpy()
        out[tag + "eps_recon"] = torch.randn([B, cfg.latent_variable_size]).numpy()
        out[tag + "enc_gen_losses"] = np.array([float(x) for x in eg])
        for n, p in cgn.named_parameters():
            if n.startswith("sentiment_discriminator.") or p.grad is None:
                continue
            out[tag + "grad/" + n] = p.grad.detach().numpy().copy()
        for k, v in cgn.state_dict().items():
            out[tag + "after/" + k] = v.detach().numpy().copy()
    hook.remove()

    # ---- validation closures on the trained-for-two-iterations model (forward only): valid_encoder_generator().valid
    # (cgn.py:1159-1250) with the z pass, discriminator_sentiment_valid().valid (:174-240) with accuracies
    cgn.prep_enc_gen_validation()
    valid = cgn.valid_encoder_generator()
    torch.manual_seed(400)
    vl = valid(batch, B, False, c_batch, 7000, calc_z_loss=True)
    torch.manual_seed(400)
    out["valid/eps"] = torch.randn([B, cfg.latent_variable_size]).numpy()
    out["valid/enc_gen_losses"] = np.array([float(x) for x in vl])
    out["valid/c"], out["valid/global_step"] = c_batch.numpy().copy(), np.array(7000)
    cgn.prep_disc_validation()
    handler.get_sentiment_dev_batch = handler.get_sentiment_train_batch
    dv = cgn.discriminator_sentiment_valid(handler, False, return_accuracy=True)(0, x_gen.tolist(), c_gen)
    out["valid/disc"] = np.array([float(x) for x in dv])  # ce labelled, ce generated, correct, generated correct

    # ---- sleep-phase beam sampler (cgn.py:747-921 + utils/beam_search.py + generator.only_decoder_beam) on the same
    # trained-for-two-iterations model. No sentence reaches <eos> at the top of its beam within seq_len here (asserted
    # below through equal lengths), so the reference's compaction of finished sentences (:843-869) - which mis-pairs
    # seeds, SURVEY.md §8f - never runs: this pins the beam arithmetic itself.
    class Vocab:
        pad_token, go_token, end_token = "_", ">", "|"
        word_to_idx = {"_": V - 3, ">": V - 2, "|": V - 1}

    S_, K_, NB_ = 3, 4, 2
    seed_zc = torch.randn(S_, cfg.latent_variable_size + 1, generator=g)
    seed_zc[:, -1] = (seed_zc[:, -1] > 0).float()
    try:
        codes, scores = cgn.sample_from_generator(Vocab(), cfg.max_seq_len, seed_zc.clone(), False, beam_size=K_,
                                                  n_best=NB_, samples=S_)
    except RuntimeError as e:
        # a sentence reached <eos>: the reference then tiles the seeds for the ORIGINAL number of sentences against
        # the compacted batch (generator.py:30-32) and its own view() fails - no sampler vectors for this case
        print("  sampler: the reference raised after a sentence finished (%s); not recorded" % str(e)[:60])
        codes = None
    if codes is not None:
        codes = np.asarray([[[int(w) for w in step] for step in sent] for sent in codes])  # [S, T, n_best]
        assert codes.shape == (S_, cfg.max_seq_len, NB_), codes.shape
        out["sampler/seed"], out["sampler/codes"] = seed_zc.numpy(), codes
        out["sampler/scores"] = np.stack([sc.detach().numpy() for sc in scores])
        out["sampler/params"] = np.array([S_, K_, NB_, cfg.max_seq_len])

    # ---- initial VAE pre-training (cgn.py:105-132, 621-674; main.py:97-118): a FRESH model from the same initial
    # state, two training steps (one Adam over every learnable parameter) and one validation step
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "word_embeddings.npy")
        np.save(path, word_embed)
        with contextlib.redirect_stdout(io.StringIO()):
            vae = cgn_mod.Controlled_Generation_Sentence(cfg, path)
    vae.load_state_dict({k[5:]: torch.from_numpy(v) for k, v in out.items() if k.startswith("init/")})

    class Loader:
        def next_batch(self, batch_size, target, start_index):
            return self.batch

    handler.gen_batch_loader = Loader()
    vae_train = vae.initial_rvae_trainer(handler)
    vae_valid = vae.initial_rvae_valid(handler)
    vae.prep_enc_gen_training()  # main.py:118
    for it in range(2):
        tag = "vae%d/" % it
        handler.gen_batch_loader.batch = [out["it%d/%s" % (it, nm)] for nm in ("enc_word", "enc_char", "dec_word")] + \
            [np.zeros((B, L + 1, cfg.max_word_len), np.int64), out["it%d/target" % it]]
        torch.manual_seed(500 + it)
        res = vae_train(3000 + 1000 * it, B, False, 0.0, 0)
        torch.manual_seed(500 + it)  # train_rvae draws eps (:96) then c ~ Bernoulli(0.5) (:119-120), in that order
        out[tag + "eps"] = torch.randn([B, cfg.latent_variable_size]).numpy()
        out[tag + "c"] = torch.bernoulli(torch.ones(B, 1) * 0.5).numpy()
        out[tag + "i"] = np.array(3000 + 1000 * it)
        out[tag + "result"] = np.array([float(x) for x in res])  # ce, kld, kld_coef(i), loss
        for n, p_ in vae.named_parameters():
            if p_.grad is not None:
                out[tag + "grad/" + n] = p_.grad.detach().numpy().copy()
        for k, v in vae.state_dict().items():
            out[tag + "after/" + k] = v.detach().numpy().copy()
    vae.prep_enc_gen_validation()  # main.py:148
    torch.manual_seed(600)
    res = vae_valid(5000, B, False, 0.0, 0)
    torch.manual_seed(600)
    out["vaev/eps"] = torch.randn([B, cfg.latent_variable_size]).numpy()
    out["vaev/c"] = torch.bernoulli(torch.ones(B, 1) * 0.5).numpy()
    out["vaev/result"] = np.array([float(x) for x in res])
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "with", len(out), "arrays,", os.path.getsize(OUT), "bytes")
    for it in range(2):
        print("it", it, "disc", out["it%d/disc_losses" % it], "enc-gen", out["it%d/enc_gen_losses" % it])


if __name__ == "__main__":
    main()
