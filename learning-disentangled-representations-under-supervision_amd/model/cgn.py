"""Controlled_Generation_Sentence — the CGN orchestrator (Hu et al. 2017) on MI355X.

Same class name, constructor, sub-module names (hence state_dict keys) and step-closure API as the reference's
model/cgn.py:24-1250. The step closures run the fused HIP step (hip/step.py); the module-level forward helpers
(encode_for_z, train_rvae, discriminator_forward) run the same kernels through the modules' autograd Functions.

The sleep-phase beam sampler (sample_from_generator, sample_generator_for_learning, sample; cgn.py:747-1034) runs
batched on the device (hip/sampler.py).
"""
import numpy as np
import torch as t
import torch.nn as nn

from ..hip import nn as hnn
from ..hip.step import CGNStep
from ..selfModules.embedding import Embedding
from ..utils.functional import kld_coef, temp_coef
from .discriminator_sentiment import Sentiment_CNN
from .encoder import Encoder
from .generator import Generator


def _to_device_long(arr, use_cuda):
    x = t.from_numpy(np.ascontiguousarray(arr)) if isinstance(arr, np.ndarray) else arr
    x = x.long()
    return x.cuda() if use_cuda else x


class Controlled_Generation_Sentence(nn.Module):

    def __init__(self, config, embedding_path):
        super().__init__()
        self.config = config
        He, Z = config.encoder_rnn_size, config.latent_variable_size
        # sub-module names are the reference's (cgn.py:26-44): they ARE the state_dict keys
        self.embedding = Embedding(config, embedding_path)
        self.encoder = Encoder(config)
        self.e2mu = hnn.Linear(2 * He, Z)
        self.e2logvar = hnn.Linear(2 * He, Z)
        self.generator = Generator(config)
        self.sentiment_discriminator = Sentiment_CNN(config)
        # Multi-attribute extension (BASELINE.json configs[4]: c = sentiment + tense, one text-CNN per attribute):
        # attribute 0 keeps the reference's module; config.c_size = 1 (the reference) leaves this list empty and the
        # state_dict unchanged.
        self.attribute_discriminators = nn.ModuleList(
            [Sentiment_CNN(config) for _ in range(getattr(config, "c_size", 1) - 1)])
        # which parameters were trainable at construction (the frozen word table is not): the *_mode() switches below
        # only ever re-enable these (the reference keeps 0/1 lists per sub-module for the same purpose, cgn.py:30-44)
        self._born_trainable = {id(p) for p in self.parameters() if p.requires_grad}

        # kld_coef variant: True reproduces the reference under its own interpreter (Python 2 floor division)
        self.kld_py2_floor = True
        # data parallelism hooks (parallel.py sets them): all-reduce callable over a flat gradient arena + world size
        self._allreduce = None
        self._world_size = 1
        self._force_dp = False
        self._step_engine = None

    # ------------------------------------------------------------------------------------------------------------
    # fused-step engine (flat parameter arenas + HIP kernels), built lazily once the model sits on the GPU
    # ------------------------------------------------------------------------------------------------------------
    def step_engine(self):
        eng = self._step_engine
        if eng is None or not eng.valid():
            eng = CGNStep(self, allreduce=self._allreduce, world_size=self._world_size, force_dp=self._force_dp)
            self._step_engine = eng
        return eng

    def set_data_parallel(self, allreduce, world_size, force=False):
        """allreduce: parallel.BucketedAllReduce-like object (begin(flat) -> handles, wait(handles)) that sums a flat
        fp32 gradient tensor in place over all ranks (RCCL all_reduce). force: keep the data-parallel schedule on for a
        one-rank group."""
        self._allreduce, self._world_size, self._force_dp = allreduce, int(world_size), bool(force)
        self._step_engine = None

    def compute_mahalabonis(self, mu1, mu2, std1, std2):
        """cgn.py:47-57 (diagnostic helper; the training step uses the fused kernel cgn_mahalanobis_f32)."""
        d = mu1 - mu2
        return (d * d / (0.5 * (std1 + std2))).sum(1).sqrt()

    def encode_for_z(self, encoder_input, batch_size, use_cuda, eps=None):
        """cgn.py:83-103. `eps` optionally replaces the t.randn draw of :96. Returns (z, std, logvar, mu)."""
        context, h_0, c_0 = self.encoder(encoder_input, None)
        mu = self.e2mu(context)
        logvar = self.e2logvar(context)
        if eps is None:
            eps = t.randn([batch_size, self.config.latent_variable_size])
        eps = eps.to(mu.device)
        z, std = _ReparamFn.apply(mu, logvar, eps)
        return z, std, logvar, mu

    def train_rvae(self, drop_prob, encoder_word_input=None, encoder_char_input=None, generator_word_input=None,
                   eps=None, c=None):
        """cgn.py:105-132: VAE forward with a random code c ~ Bernoulli(0.5). Returns (out, final_state, kld, mu, std, z)."""
        use_cuda = self.embedding.word_embed.weight.is_cuda
        [batch_size, _] = encoder_word_input.size()
        encoder_input = self.embedding(encoder_word_input, encoder_char_input)
        z, std, logvar, mu = self.encode_for_z(encoder_input, batch_size, use_cuda, eps)
        if c is None:
            c = t.bernoulli(t.ones(batch_size, self.config.c_size) * 0.5)
        c = c.to(z.device).view(batch_size, -1)
        input_code = _ZCFn.apply(z, c)
        kld = _KldFn.apply(mu, logvar)
        generator_input = self.embedding.word_embed(generator_word_input)
        out, final_state = self.generator(generator_input, input_code, drop_prob, None)
        return out, final_state, kld, mu, std, z

    def discriminator_forward(self, input_data, batch_size, use_cuda, uniform=None):
        """cgn.py:134-172: c ~ Bernoulli(softmax(D(word_embed(encoder_word_input)))[:, 1]); returns c [B] (float).
        `uniform` optionally supplies the U[0,1) draw behind the Bernoulli sample."""
        self.sentiment_discriminator.eval()
        encoder_word_input = _to_device_long(input_data[0], use_cuda)
        eng = self.step_engine()
        if uniform is None:
            C_ = getattr(self.config, "c_size", 1)
            uniform = t.rand(encoder_word_input.size(0)) if C_ == 1 else t.rand(encoder_word_input.size(0), C_)
        c, _ = eng.infer_c(encoder_word_input.contiguous(), uniform.to(eng.dev, t.float32).contiguous())
        return c

    # ------------------------------------------------------------------------------------------------------------
    # discriminator step closures (cgn.py:174-343)
    # ------------------------------------------------------------------------------------------------------------
    def _disc_batches(self, data_handler, batch_X, batch_Y, generated_samples, generated_c, use_cuda):
        dev_long = lambda x: _to_device_long(x, use_cuda).contiguous()  # noqa: E731
        X = dev_long(data_handler.create_sentiment_batch(batch_X))
        Y = dev_long(np.array(batch_Y))
        GX = dev_long(data_handler.create_sentiment_batch(generated_samples))
        C_ = getattr(self.config, "c_size", 1)
        GY = generated_c.view(-1).long() if C_ == 1 else generated_c.view(-1, C_).long()
        GY = (GY.cuda() if use_cuda else GY).contiguous()
        return X, Y, GX, GY

    def _disc_masks(self, B_l, B_g, dev):
        p = self.config.sentiment_dropout
        if not self.sentiment_discriminator.training or not p:
            return None, None
        F_ = len(self.config.sentiment_kernel_size) * self.config.sentiment_kernel_num
        keep = 1.0 - p
        return ((t.rand(B_l, F_, device=dev) < keep).float() / keep, (t.rand(B_g, F_, device=dev) < keep).float() / keep)

    def discriminator_sentiment_valid(self, data_handler, use_cuda, return_accuracy=False):
        """cgn.py:174-240."""

        def valid(batch_index, generated_samples, generated_c):
            self.sentiment_discriminator.eval()
            batch_X, batch_Y = data_handler.get_sentiment_dev_batch(batch_index)
            X, Y, GX, GY = self._disc_batches(data_handler, batch_X, batch_Y, generated_samples, generated_c, use_cuda)
            eng = self.step_engine()
            _, ce_l, _, ce_g, _ = eng.disc_train(None, X, Y, GX, GY, train=False)
            if not return_accuracy:
                return ce_l, ce_g
            C_ = getattr(self.config, "c_size", 1)
            half = lambda n: t.full((n,) if C_ == 1 else (n, C_), 0.5, device=eng.dev)  # noqa: E731
            c_l, p_l = eng.infer_c(X, half(X.size(0)))   # c = [p1 > 0.5] = argmax
            c_g, p_g = eng.infer_c(GX, half(GX.size(0)))
            correct = (c_l.long() == Y).sum()
            generated_correct = (c_g.long() == GY).sum()
            return ce_l, ce_g, correct, generated_correct

        return valid

    def discriminator_sentiment_trainer(self, data_handler, use_cuda):
        """cgn.py:242-343: returns train(generated_samples, generated_c, batch_index) -> 5-tuple
        (total_loss, batch_train_ce_loss, generated_loss, generated_loss_ce, emperical_shannon_entropy).
        Extra keyword arguments mask_labelled / mask_generated pin the dropout masks (parity tests)."""
        optimizer = [None]

        def train(generated_samples, generated_c, batch_index, mask_labelled=None, mask_generated=None, capture=None):
            batch_X, batch_Y = data_handler.get_sentiment_train_batch(batch_index)
            X, Y, GX, GY = self._disc_batches(data_handler, batch_X, batch_Y, generated_samples, generated_c, use_cuda)
            eng = self.step_engine()
            if optimizer[0] is None or optimizer[0].arena is not eng.arenas["disc"]:
                optimizer[0] = eng.new_optimizer("disc")
            if mask_labelled is None and mask_generated is None:
                mask_labelled, mask_generated = self._disc_masks(X.size(0), GX.size(0), eng.dev)
            return eng.disc_train(optimizer[0], X, Y, GX, GY, mask_labelled, mask_generated, capture=capture)

        return train

    # ------------------------------------------------------------------------------------------------------------
    # mode switches (cgn.py:346-395, 688-745) — kept for API compatibility; the fused step decides what to update from
    # the closure being called, exactly as the reference's optimizers do from their parameter lists
    # ------------------------------------------------------------------------------------------------------------
    def _set_phase(self, train_disc, train_enc_gen):
        """requires_grad + train()/eval() of the three parts in one place (cgn.py:346-395 spell out the four cases)."""
        self.discriminator_mode(train_mode=train_disc)
        self.encoder_mode(train_mode=train_enc_gen)
        self.generator_mode(train_mode=train_enc_gen)
        for d in self._discriminators():
            d.train(train_disc)
        self.encoder.train(train_enc_gen)
        self.generator.train(train_enc_gen)

    def prep_disc_validation(self):
        self._set_phase(False, False)

    def prep_disc_training(self):
        self._set_phase(True, False)

    def prep_enc_gen_validation(self):
        self._set_phase(False, False)

    def prep_enc_gen_training(self):
        self._set_phase(False, True)

    def _discriminators(self):
        return [self.sentiment_discriminator] + list(self.attribute_discriminators)

    @staticmethod
    def _trainable(*modules):
        return [p for m in modules for p in m.parameters() if p.requires_grad]

    def encoder_params(self):
        """cgn.py:676-683: the encoder optimizer's parameters, in the reference's order."""
        return self._trainable(self.encoder, self.embedding, self.e2mu, self.e2logvar)

    def generator_params(self):
        return self._trainable(self.generator)

    def sentiment_discriminator_parameters(self):
        return self._trainable(*self._discriminators())

    def learnable_parameters(self):
        return self._trainable(self)

    def _switch(self, modules, on):
        for m in modules:
            for p in m.parameters():
                p.requires_grad = bool(on) and id(p) in self._born_trainable

    def encoder_mode(self, train_mode=False):
        self._switch((self.encoder, self.embedding, self.e2mu, self.e2logvar), train_mode)

    def discriminator_mode(self, train_mode=True):
        self._switch(self._discriminators(), train_mode)

    def generator_mode(self, train_mode=True):
        self._switch((self.generator,), train_mode)

    # ------------------------------------------------------------------------------------------------------------
    # initial VAE pre-training (cgn.py:621-674): one optimizer over every learnable parameter, no detach, no D
    # ------------------------------------------------------------------------------------------------------------
    def _rvae_step(self, opts, input, i, dropout, train, eps=None, c=None, capture=None):
        eng = self.step_engine()
        use_cuda = True
        enc_word, enc_char, dec_word, _, target = [None if v is None else _to_device_long(v, use_cuda).contiguous()
                                                   for v in input]
        B = enc_word.size(0)
        if eps is None:
            eps = t.randn([B, self.config.latent_variable_size])
        if c is None:
            c = t.bernoulli(t.ones(B, self.config.c_size) * 0.5)  # cgn.py:119-120
        mask = _gen_mask(self.generator, dec_word, dropout, self.config.word_embed_size, eng.dev) if train else None
        kc = kld_coef(i, py2_floor=self.kld_py2_floor)
        ce, kld, _, _, enc_loss, _ = eng.enc_gen_train(
            opts, enc_word, enc_char, dec_word, target, c.to(eng.dev, t.float32).contiguous(),
            eps.to(eng.dev, t.float32).contiguous(), kc, 1.0, gen_drop_mask=mask, calc_z_loss=False, train=train,
            use_disc=False, joint=True, capture=capture)
        return ce, kld, kc, enc_loss  # loss = 79*ce + kld_coef*kld (cgn.py:641)

    def initial_rvae_trainer(self, data_handler):
        opts = [None]

        def train(i, batch_size, use_cuda, dropout, start_index, eps=None, c=None, capture=None):
            input = data_handler.gen_batch_loader.next_batch(batch_size, 'train', start_index)
            eng = self.step_engine()
            if opts[0] is None or opts[0][0].arena is not eng.arenas["enc"]:
                opts[0] = (eng.new_optimizer("enc"), eng.new_optimizer("gen"))
            return self._rvae_step(opts[0], input, i, dropout, True, eps, c, capture)

        return train

    def initial_rvae_valid(self, data_handler):
        def valid(i, batch_size, use_cuda, dropout, start_index, eps=None, c=None):
            input = data_handler.gen_batch_loader.next_batch(batch_size, 'valid', start_index)
            return self._rvae_step(None, input, i, 0, False, eps, c)

        return valid

    # ------------------------------------------------------------------------------------------------------------
    # encoder/generator step closures (cgn.py:1043-1250)
    # ------------------------------------------------------------------------------------------------------------
    def _enc_gen(self, opts, input, use_cuda, dropout, c_target, global_step, calc_z_loss, train, eps, gen_drop_mask,
                 capture):
        if not use_cuda:
            raise RuntimeError("the CGN step runs on the MI355X only (use_cuda must be True); there is no CPU path")
        enc_word, enc_char, dec_word, _, target = [None if v is None else _to_device_long(v, use_cuda).contiguous()
                                                   for v in input]
        eng = self.step_engine()
        B = enc_word.size(0)
        if eps is None:
            eps = t.randn([B, self.config.latent_variable_size])  # cgn.py:96 (CPU draw, then moved)
        if train and gen_drop_mask is None:
            gen_drop_mask = _gen_mask(self.generator, dec_word, dropout, self.config.word_embed_size, eng.dev)
        kld_coeff = kld_coef(global_step, py2_floor=self.kld_py2_floor)  # cgn.py:1074
        logit_temperature = temp_coef(global_step)  # cgn.py:1086
        c = c_target.to(eng.dev, t.float32).contiguous()
        ce, kld, z_loss, c_loss, enc_loss, gen_loss = eng.enc_gen_train(
            opts, enc_word, enc_char, dec_word, target, c, eps.to(eng.dev, t.float32).contiguous(), kld_coeff,
            logit_temperature, gen_drop_mask=gen_drop_mask, calc_z_loss=calc_z_loss, capture=capture, train=train)
        return ce, kld, z_loss, c_loss, kld_coeff, logit_temperature, enc_loss, gen_loss

    def train_encoder_generator(self, summary_writer=None):
        """cgn.py:1043-1156. Returns train(input, batch_size, use_cuda, dropout, c_target, global_step,
        calc_z_loss=True) -> (cross_entropy, kld, z_loss, c_loss, kld_coeff, logit_temperature, total_encoder_loss,
        total_generator_loss). Keyword-only extras: eps, gen_drop_mask (explicit random draws), capture (dict that
        receives g and the gradients seen by each optimizer)."""
        opts = [None]

        def train(input, batch_size, use_cuda, dropout, c_target, global_step, calc_z_loss=True, *, eps=None,
                  gen_drop_mask=None, capture=None):
            eng = self.step_engine()
            if opts[0] is None or opts[0][0].arena is not eng.arenas["enc"]:
                opts[0] = (eng.new_optimizer("enc"), eng.new_optimizer("gen"))  # cgn.py:1046-1047
            return self._enc_gen(opts[0], input, use_cuda, dropout, c_target, global_step, calc_z_loss, True, eps,
                                 gen_drop_mask, capture)

        return train

    def valid_encoder_generator(self):
        """cgn.py:1159-1250 (forward only, dropout 0)."""

        def valid(input, batch_size, use_cuda, c_target, global_step, calc_z_loss=True, *, eps=None):
            return self._enc_gen(None, input, use_cuda, 0.0, c_target, global_step, calc_z_loss, False, eps, None, None)

        return valid

    # ------------------------------------------------------------------------------------------------------------
    # ------------------------------------------------------------------------------------------------------------
    # sleep-phase sampler (cgn.py:747-1034): batched beam search on the device (hip/sampler.py)
    # ------------------------------------------------------------------------------------------------------------
    def sample_from_generator(self, batch_loader, seq_len, seed, use_cuda, beam_size=10, n_best=1, samples=5,
                              learning=False):
        """cgn.py:747-921. Returns (all_sent_codes, allScores): all_sent_codes[b] = list over time of n_best-tuples of
        word ids (the zip(*hyps) of :879), allScores[b] = the n_best best joint log-probabilities.
        learning=True (:880,901-902): returns all_sent_probs instead: all_sent_probs[b] = the [len_b, V] device tensor of
        log_softmax rows along the best hypothesis (Beam.get_hyp_probs, beam_search.py:140-152)."""
        if not use_cuda:
            raise RuntimeError("the sampler runs on the MI355X only (use_cuda must be True)")
        self.generator.eval()
        eng = self.step_engine()
        from ..hip.sampler import BeamSampler
        w2i = batch_loader.word_to_idx
        go, eos, pad = w2i[batch_loader.go_token], w2i[batch_loader.end_token], w2i[batch_loader.pad_token]
        seed = seed[:samples].to(eng.dev, t.float32).contiguous()
        if learning:
            return BeamSampler(eng).sample(seed, seq_len, beam_size, go, eos, pad, n_best=n_best, learning=True)
        hyps, scores = BeamSampler(eng).sample(seed, seq_len, beam_size, go, eos, pad, n_best=n_best)
        all_sent_codes = [list(zip(*h)) for h in hyps]
        allScores = [sc[:n_best] for sc in scores]
        return all_sent_codes, allScores

    def sample_generator_for_learning(self, data_handler, config, use_cuda=True, sample=10, seed_z=None, seed_c=None):
        """cgn.py:993-1034: `sample` sentences from z ~ N(0,I), c ~ Bernoulli(0.5). Returns (list of word-id lists,
        seed_c [sample,1]). seed_z / seed_c optionally pin the random draws."""
        self.generator.eval()
        if seed_z is None:
            seed_z = t.randn([sample, config.latent_variable_size])
        if seed_c is None:
            seed_c = t.bernoulli(t.ones(sample, 1) * 0.5)
        seed = t.cat((seed_z, seed_c), 1)
        generated_samples, _ = self.sample_from_generator(data_handler.gen_batch_loader, config.max_seq_len, seed,
                                                          use_cuda, n_best=1, samples=sample, learning=False)
        all_gen_samples = [[word_n[0] for word_n in batch] for batch in generated_samples]
        return all_gen_samples, seed_c

    def sample(self, data_handler, config, use_cuda=True, print_sentences=False, samp=5, seed_z=None):
        """cgn.py:923-991: `samp` latent draws, each decoded with c = 1 and c = 0. Returns (sentences, seed_c)."""
        if seed_z is None:
            seed_z = t.randn([samp, config.latent_variable_size])
        seed_z = seed_z.unsqueeze(1).repeat(1, 2, 1).view(-1, config.latent_variable_size)
        seed_c = t.ones(2 * samp, 1)
        seed_c[1::2] = 0
        seed = t.cat((seed_z, seed_c), 1)
        sentences, _ = self.sample_from_generator(data_handler.gen_batch_loader, config.max_seq_len, seed, use_cuda,
                                                  n_best=1, samples=2 * samp)
        all_sentences = [[word_n[0] for word_n in batch] for batch in sentences]
        if print_sentences:
            loader = data_handler.gen_batch_loader
            for c, sen in zip(seed_c.view(-1).tolist(), all_sentences):
                print('c={%d} : %s' % (c, ' '.join(loader.decode_word(w) for w in sen)))
        return all_sentences, seed_c


def _gen_mask(generator, dec_word, dropout, E, dev):
    """Word-dropout mask for F.dropout(generator_input, drop_prob) (generator.py:52); see generator._dropout_mask."""
    if isinstance(dropout, t.Tensor):
        return dropout.to(dev, t.float32).contiguous()
    if generator.legacy_dropout_noop or not dropout:
        return None
    keep = 1.0 - float(dropout)
    B, T = dec_word.shape
    return (t.rand(B, T, E, device=dev) < keep).float() / keep


class _ReparamFn(t.autograd.Function):
    """z = eps * exp(0.5 * logvar) + mu ; std = exp(0.5 * logvar)   (cgn.py:94-101)."""

    @staticmethod
    def forward(ctx, mu, logvar, eps):
        from ..hip import ops
        B, Z = mu.shape
        mulv = hnn.new((B, 2 * Z), mu.device)
        ops.copy2d(hnn.f32c(mu), mulv[:, :Z])
        ops.copy2d(hnn.f32c(logvar), mulv[:, Z:])
        zc, std, kld = hnn.new((B, Z), mu.device), hnn.new((B, Z), mu.device), hnn.new((1,), mu.device)
        e = hnn.f32c(eps.to(t.float32))
        ops.reparam_kld_fwd(mulv, e, None, zc, std, kld)
        ctx.save_for_backward(mulv, e)
        return zc, std

    @staticmethod
    def backward(ctx, dz, dstd):
        from ..hip import ops
        mulv, e = ctx.saved_tensors
        B, Z2 = mulv.shape
        Z = Z2 // 2
        d = hnn.new((B, Z2), mulv.device)
        ops.reparam_kld_bwd(mulv, e, hnn.f32c(dz), 0.0, d)
        dmu, dlv = hnn.f32c(d[:, :Z]), hnn.f32c(d[:, Z:])
        if dstd is not None:
            # d std / d logvar = 0.5 * std ; fold through the same kernel with eps := 1
            ones = hnn.new((B, Z), mulv.device)
            ops.fill(ones, 1.0)
            d2 = hnn.new((B, Z2), mulv.device)
            ops.reparam_kld_bwd(mulv, ones, hnn.f32c(dstd), 0.0, d2)
            ops.copy2d(d2[:, Z:], dlv, beta=1.0)
        return dmu, dlv, None


class _KldFn(t.autograd.Function):
    """kld = mean_b(-0.5 * sum_z(logvar - mu^2 - exp(logvar) + 1))   (cgn.py:127, :1076)."""

    @staticmethod
    def forward(ctx, mu, logvar):
        from ..hip import ops
        B, Z = mu.shape
        mulv = hnn.new((B, 2 * Z), mu.device)
        ops.copy2d(hnn.f32c(mu), mulv[:, :Z])
        ops.copy2d(hnn.f32c(logvar), mulv[:, Z:])
        eps = hnn.new((B, Z), mu.device)
        ops.fill(eps, 0.0)
        zc, std, kld = hnn.new((B, Z), mu.device), hnn.new((B, Z), mu.device), hnn.new((1,), mu.device)
        ops.reparam_kld_fwd(mulv, eps, None, zc, std, kld)
        ctx.save_for_backward(mulv, eps)
        return kld.view(())

    @staticmethod
    def backward(ctx, dk):
        from ..hip import ops
        mulv, eps = ctx.saved_tensors
        B, Z2 = mulv.shape
        Z = Z2 // 2
        d = hnn.new((B, Z2), mulv.device)
        ops.reparam_kld_bwd(mulv, eps, None, float(dk) / B, d)
        return hnn.f32c(d[:, :Z]), hnn.f32c(d[:, Z:])


class _ZCFn(t.autograd.Function):
    """t.cat((z, c), 1)   (cgn.py:125, :1080)."""

    @staticmethod
    def forward(ctx, z, c):
        from ..hip import ops
        B, Z = z.shape
        out = hnn.new((B, Z + c.shape[1]), z.device)
        ops.copy2d(hnn.f32c(z), out[:, :Z])
        ops.copy2d(hnn.f32c(c.to(t.float32)), out[:, Z:])
        ctx.Z = Z
        return out

    @staticmethod
    def backward(ctx, d):
        return hnn.f32c(hnn.f32c(d)[:, :ctx.Z]), None
