"""CPU ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product path).

A restatement, in modern PyTorch (CPU, fp32 or fp64), of the reference's CGN wake-sleep training step
(nikhil-dce/Learning-Disentangled-Representations-under-Supervision). Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this file, and only as the checker / reported CPU baseline.

Every function cites the reference file:line it follows. Differences from the reference source are limited to
  * Python 3 / torch 2.x spelling (retain_variables -> retain_graph, explicit softmax dims, keepdim-era squeeze),
  * every random draw is an explicit argument (eps, c, dropout masks, bernoulli uniforms) so that the HIP path can be
    compared on identical inputs,
  * hyper-parameters are parametric (OracleConfig) with the reference's values as defaults.

Pinning: the reference has no tests, golden vectors or fixtures. This oracle is pinned by
outputs of the reference's own code run in the build container and committed as data under tests/golden/:
  * module level (tests/golden/make_golden.py -> reference_modules.npz, checked by tests/test_oracle_golden.py):
    Generator, Highway, Encoder, Sentiment_CNN forward outputs, kld_coef / temp_coef, Config facts;
  * STEP level (tests/golden/make_golden_step.py -> reference_step.npz, checked by tests/test_oracle_step_golden.py):
    the reference's own closures - discriminator_sentiment_trainer().train (cgn.py:269-341), discriminator_forward
    (:134-172) and train_encoder_generator().train (:1049-1154) - executed for two consecutive wake-sleep iterations
    (Python-2 files converted in memory with lib2to3; two textual substitutions for renamed / removed torch spellings,
    listed in that script). Every loss, every gradient the closures leave in .grad (incl. the z.grad alias routing)
    and every parameter after Adam agree with this file to fp32 rounding (most of them bit for bit).
    The same fixture holds the reference's beam sampler run (sample_from_generator + Beam + only_decoder_beam): this
    file's sample_from_generator reproduces its token sequences and scores.
The multi-attribute extension (c_size > 1) has no counterpart in the reference: parity unpinned for it.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------------------------
# utils/config.py:4-37 (parametric; defaults = reference values)
# --------------------------------------------------------------------------------------------------------------
class OracleConfig:
    def __init__(self, max_word_len=12, max_seq_len=15, word_vocab_size=1000, char_vocab_size=64,
                 learning_rate=5e-5, lambda_c=7.9, lambda_z=0.1, lambda_u=0.1, beta=1.0,
                 word_embed_size=300, char_embed_size=15,
                 kernels=((1, 25), (2, 50), (3, 75), (4, 100), (5, 125), (6, 150)),
                 encoder_rnn_size=600, encoder_num_layers=1, latent_variable_size=1100,
                 decoder_rnn_size=800, decoder_num_layers=2,
                 sentiment_kernel_size=(3, 4, 5), sentiment_kernel_num=100, sentiment_dropout=0.3, c_size=1):
        self.learning_rate = learning_rate
        self.max_word_len = int(max_word_len)
        self.max_seq_len = int(max_seq_len) + 1  # config.py:9 (go or eos token)
        self.word_vocab_size = int(word_vocab_size)
        self.char_vocab_size = int(char_vocab_size)
        self.word_embed_size = word_embed_size
        self.char_embed_size = char_embed_size
        self.kernels = [tuple(k) for k in kernels]
        self.sum_depth = sum(d for _, d in self.kernels)  # config.py:18
        self.encoder_rnn_size = encoder_rnn_size
        self.encoder_num_layers = encoder_num_layers
        self.latent_variable_size = latent_variable_size
        self.decoder_rnn_size = decoder_rnn_size
        self.decoder_num_layers = decoder_num_layers
        self.sentiment_kernel_size = list(sentiment_kernel_size)
        self.sentiment_kernel_num = sentiment_kernel_num
        self.sentiment_dropout = sentiment_dropout
        self.lambda_c, self.lambda_u, self.lambda_z, self.beta = lambda_c, lambda_u, lambda_z, beta
        self.c_size = c_size  # the reference hard-codes 1 (generator.py:17,54)


# --------------------------------------------------------------------------------------------------------------
# utils/functional.py:28-42
# --------------------------------------------------------------------------------------------------------------
def kld_coef(i, extended=False, py2_floor=True):
    """functional.py:28-33. The reference is Python 2: with an int step, (i - 3500)/1000 is FLOOR division.
    py2_floor=True (default) reproduces that; False gives the Python-3 reading of the same source line."""
    if not extended:
        x = (i - 3500) // 1000 if (py2_floor and isinstance(i, (int, np.integer))) else (i - 3500) / 1000
    else:
        x = (i - 50000) // 15000 if (py2_floor and isinstance(i, (int, np.integer))) else (i - 50000) / 15000
    return (math.tanh(x) + 1) / 2


def temp_coef(i):
    """functional.py:35-42."""
    return float(np.maximum(1.0 * np.exp(-0.00002 * i), 0.1))


# --------------------------------------------------------------------------------------------------------------
# Test aid (no counterpart in the reference): which side of each non-differentiable point the forward pass takes.
# The step is piecewise smooth: ReLU gates (highway.py:39, discriminator_sentiment.py:64) and max-over-time winners
# (tdnn.py:43, discriminator_sentiment.py:66). Where two candidates are tied to within fp32 rounding, an fp32
# implementation may legitimately pick the other one, and the GRADIENT (not the loss) then differs by that gate's
# whole contribution. While SELECT is a Selections object the modules below report every such choice to it; given the
# choices an fp32 run made (`forced`), the oracle adopts them where - and only where - its own candidates are tied to
# within `tie` (relative), and counts the adoptions. With SELECT = None (default) the plain torch ops run.
# --------------------------------------------------------------------------------------------------------------
class Selections:
    def __init__(self, forced=None, tie=1e-5):
        self.forced = forced or {}
        self.tie = tie
        self.scope = ""
        self.adopted = {}   # site -> number of near-tied choices taken over from `forced`
        self.refused = {}   # site -> number of differing choices that were NOT near ties (kept the oracle's own)
        self.seen = []

    def _count(self, site, take, differ):
        n_take, n_ref = int(take.sum()), int((differ & ~take).sum())
        if n_take:
            self.adopted[site] = self.adopted.get(site, 0) + n_take
        if n_ref:
            self.refused[site] = self.refused.get(site, 0) + n_ref

    def relu(self, name, pre):
        site = self.scope + "/" + name
        self.seen.append(site)
        own = pre.detach() > 0
        f = self.forced.get(site)
        if f is not None:
            f = f.to(own.device).bool()
            differ = own != f
            take = differ & (pre.detach().abs() < self.tie * pre.detach().abs().mean())
            self._count(site, take, differ)
            own = torch.where(take, f, own)
        return pre * own.to(pre.dtype)

    def maxpool(self, name, v, relu):
        """v [N, C, P] -> [N, C]: max over P (then ReLU if `relu`); forced: [N, C] winner positions, -1 = gate closed."""
        site = self.scope + "/" + name
        self.seen.append(site)
        d = v.detach()
        thr = self.tie * d.abs().mean()
        idx = d.argmax(2)
        f = self.forced.get(site)
        if f is not None:
            f = f.to(idx.device).long()
            fpos = f.clamp(min=0)
            differ = (f >= 0) & (fpos != idx)
            gap = d.gather(2, idx.unsqueeze(2)).squeeze(2) - d.gather(2, fpos.unsqueeze(2)).squeeze(2)
            take = differ & (gap < thr)
            self._count(site + ":winner", take, differ)
            idx = torch.where(take, fpos, idx)
        val = v.gather(2, idx.unsqueeze(2)).squeeze(2)
        if not relu:
            return val
        is_open = val.detach() > 0
        if f is not None:
            differ = is_open != (f >= 0)
            take = differ & (val.detach().abs() < thr)
            self._count(site + ":gate", take, differ)
            is_open = torch.where(take, f >= 0, is_open)
        return val * is_open.to(val.dtype)


SELECT = None


def _scope(name):
    if SELECT is not None:
        SELECT.scope = name


# --------------------------------------------------------------------------------------------------------------
# selfModules/tdnn.py:8-52
# --------------------------------------------------------------------------------------------------------------
class TDNN(nn.Module):
    def __init__(self, params):
        super().__init__()
        self.params = params
        self.kernels = [nn.Parameter(torch.Tensor(od, params.char_embed_size, kW).uniform_(-1, 1))
                        for kW, od in params.kernels]  # tdnn.py:14-15
        for i, p in enumerate(self.kernels):
            self.register_parameter("TDNN_kernel-%d" % i, p)  # tdnn.py:50-52

    def forward(self, x):
        B, L, W, Cc = x.shape  # tdnn.py:34
        x = x.reshape(-1, self.params.max_word_len, self.params.char_embed_size).transpose(1, 2).contiguous()  # :40
        xs = [torch.tanh(F.conv1d(x, k)) for k in self.kernels]  # :42 (no bias, stride 1, no pad)
        if SELECT is not None:
            xs = [SELECT.maxpool("tdnn%d" % i, v, relu=False) for i, v in enumerate(xs)]
        else:
            xs = [v.max(2)[0] for v in xs]  # :43 (keepdim-era squeeze(2) dropped)
        return torch.cat(xs, 1).view(B, L, -1)  # :45-46


# --------------------------------------------------------------------------------------------------------------
# selfModules/embedding.py:10-74
# --------------------------------------------------------------------------------------------------------------
class Embedding(nn.Module):
    def __init__(self, params, word_embed):
        """word_embed: numpy [V,E] (the reference np.load()s it from a path, embedding.py:16)."""
        super().__init__()
        self.params = params
        if isinstance(word_embed, str):
            word_embed = np.load(word_embed)
        self.word_embed = nn.Embedding(params.word_vocab_size, params.word_embed_size)
        self.char_embed = nn.Embedding(params.char_vocab_size, params.char_embed_size)
        self.word_embed.weight = nn.Parameter(torch.from_numpy(np.asarray(word_embed)).float(), requires_grad=False)
        self.char_embed.weight = nn.Parameter(
            torch.Tensor(params.char_vocab_size, params.char_embed_size).uniform_(-1, 1))  # :24-25
        self.TDNN = TDNN(params)

    def forward(self, word_input, character_input):  # embedding.py:29-55
        B, L = word_input.shape
        w = self.word_embed(word_input)
        c = self.get_character_embed(character_input)
        return torch.cat([w, c], 2)

    def get_word_embed(self, word_input):  # :57
        return self.word_embed(word_input)

    def get_character_embed(self, character_input):  # :60-74
        B, L, _ = character_input.shape
        c = self.char_embed(character_input.reshape(-1, self.params.max_word_len))
        c = c.view(B, L, self.params.max_word_len, self.params.char_embed_size)
        return self.TDNN(c)


# --------------------------------------------------------------------------------------------------------------
# selfModules/highway.py:6-50
# --------------------------------------------------------------------------------------------------------------
class Highway(nn.Module):
    def __init__(self, size, num_layers, f):
        super().__init__()
        self.num_layers = num_layers
        self.f = f
        for kind in ("nonlinear", "linear", "gate"):  # highway.py:13-23: same creation + registration order
            mods = [nn.Linear(size, size) for _ in range(num_layers)]
            setattr(self, "_" + kind, mods)
            for i, m in enumerate(mods):
                for j, p in enumerate(m.parameters()):
                    self.register_parameter("%s_module_%d-%d" % (kind, i, j), p)

    def forward(self, x):  # highway.py:27-46
        for l in range(self.num_layers):
            gate = torch.sigmoid(self._gate[l](x))
            if SELECT is not None and self.f is F.relu:
                nonlinear = SELECT.relu("hw%d" % l, self._nonlinear[l](x))
            else:
                nonlinear = self.f(self._nonlinear[l](x))
            linear = self._linear[l](x)
            x = gate * nonlinear + (1 - gate) * linear
        return x


# --------------------------------------------------------------------------------------------------------------
# model/encoder.py:9-48
# --------------------------------------------------------------------------------------------------------------
class Encoder(nn.Module):
    def __init__(self, params):
        super().__init__()
        self.params = params
        din = params.word_embed_size + params.sum_depth
        self.hw1 = Highway(din, 2, F.relu)  # encoder.py:15
        self.rnn = nn.LSTM(input_size=din, hidden_size=params.encoder_rnn_size,
                           num_layers=params.encoder_num_layers, batch_first=True, bidirectional=True)  # :17-21

    def forward(self, input, State=None):
        B, L, D = input.shape
        x = self.hw1(input.reshape(-1, D)).view(B, L, D)  # :31-33
        _, (h_n, c_n) = self.rnn(x, State)  # :40 — NB the reference names c_n "final_state"
        fs = c_n.view(self.params.encoder_num_layers, 2, B, self.params.encoder_rnn_size)[-1]  # :43-44
        ctx = torch.cat([fs[0], fs[1]], 1)  # :45-46 (concat of final CELL states)
        return ctx, h_n, c_n  # :48


# --------------------------------------------------------------------------------------------------------------
# model/generator.py:10-64 (forward only; only_decoder_beam belongs to the sampler)
# --------------------------------------------------------------------------------------------------------------
class Generator(nn.Module):
    def __init__(self, params):
        super().__init__()
        self.params = params
        self.rnn = nn.LSTM(input_size=params.latent_variable_size + params.c_size + params.word_embed_size,
                           hidden_size=params.decoder_rnn_size, num_layers=params.decoder_num_layers,
                           batch_first=True)  # :17-20
        self.fc = nn.Linear(params.decoder_rnn_size, params.word_vocab_size)  # :22

    def forward(self, generator_input, zc, drop_mask=None, initial_state=None):
        """generator.py:45-64. `drop_mask` replaces F.dropout(generator_input, drop_prob) (:52): it is the explicit,
        already 1/(1-p)-scaled mask; None = no dropout (what the reference's torch<=0.3 F.dropout default did)."""
        B, T, _ = generator_input.shape
        if drop_mask is not None:
            generator_input = generator_input * drop_mask
        zc_t = torch.cat([zc] * T, 1).view(B, T, -1)  # :54
        x = torch.cat((generator_input, zc_t), 2)  # :56
        out, final_state = self.rnn(x, initial_state)  # :58
        out = out.contiguous().view(-1, self.params.decoder_rnn_size)
        logits = self.fc(out).view(B, T, self.params.word_vocab_size)  # :61-62
        return logits, final_state


# --------------------------------------------------------------------------------------------------------------
# model/discriminator_sentiment.py:11-78
# --------------------------------------------------------------------------------------------------------------
class Sentiment_CNN(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        D, Co, Ks = config.word_embed_size, config.sentiment_kernel_num, config.sentiment_kernel_size
        self.convs1 = nn.ModuleList([nn.Conv2d(1, Co, (K, D)) for K in Ks])  # :41
        self.fc1 = nn.Linear(len(Ks) * Co, 2)  # :49

    def features(self, x):
        x = x.unsqueeze(1)  # :62
        if SELECT is not None:  # max_t relu(v) = relu(max_t v)
            xs = [SELECT.maxpool("conv%d" % i, conv(x).squeeze(3), relu=True) for i, conv in enumerate(self.convs1)]
            return torch.cat(xs, 1)
        xs = [F.relu(conv(x)).squeeze(3) for conv in self.convs1]  # :64
        xs = [F.max_pool1d(i, i.size(2)).squeeze(2) for i in xs]  # :66
        return torch.cat(xs, 1)  # :68

    def forward(self, x, drop_mask=None):
        """:56-78. drop_mask replaces nn.Dropout (:76): explicit 1/(1-p)-scaled mask [B, |Ks|*Co]; None = eval mode."""
        f = self.features(x)
        if drop_mask is not None:
            f = f * drop_mask
        return self.fc1(f)  # :77


# --------------------------------------------------------------------------------------------------------------
# model/cgn.py:24-57, 83-103
# --------------------------------------------------------------------------------------------------------------
class CGN(nn.Module):
    """Controlled_Generation_Sentence (cgn.py:24-44): same sub-module names, hence the same state_dict keys."""

    def __init__(self, config, word_embed):
        super().__init__()
        self.config = config
        self.embedding = Embedding(config, word_embed)
        self.encoder = Encoder(config)
        self.e2mu = nn.Linear(config.encoder_rnn_size * 2, config.latent_variable_size)  # :37
        self.e2logvar = nn.Linear(config.encoder_rnn_size * 2, config.latent_variable_size)  # :38
        self.generator = Generator(config)
        self.sentiment_discriminator = Sentiment_CNN(config)
        # Multi-attribute extension (BASELINE.json configs[4]: c = sentiment + tense, one text-CNN per attribute, after
        # Hu et al. 2017 §3): attribute 0 keeps the reference's module name; the reference itself has c_size = 1, for
        # which this list is empty and the state_dict is unchanged. No counterpart in the reference: parity unpinned.
        self.attribute_discriminators = nn.ModuleList([Sentiment_CNN(config) for _ in range(config.c_size - 1)])

    def discriminators(self):
        return [self.sentiment_discriminator] + list(self.attribute_discriminators)

    # cgn.py:676-686, 1036-1037
    def encoder_params(self):
        ps = [p for p in self.encoder.parameters() if p.requires_grad]
        ps += [p for p in self.embedding.parameters() if p.requires_grad]
        ps += list(self.e2mu.parameters()) + list(self.e2logvar.parameters())
        return ps

    def generator_params(self):
        return [p for p in self.generator.parameters() if p.requires_grad]

    def sentiment_discriminator_parameters(self):
        return [p for d in self.discriminators() for p in d.parameters() if p.requires_grad]

    def encode_for_z(self, encoder_input, eps):
        """cgn.py:83-103 with the t.randn draw (:96) passed in as `eps`."""
        context, h_0, c_0 = self.encoder(encoder_input, None)
        mu = self.e2mu(context)
        logvar = self.e2logvar(context)
        std = torch.exp(0.5 * logvar)
        z = eps * std + mu
        return z, std, logvar, mu


def compute_mahalabonis(mu1, mu2, std1, std2):
    """cgn.py:47-57."""
    std = (std1 + std2) / 2
    delta_mu = mu1 - mu2
    return torch.sqrt(torch.sum(delta_mu * torch.reciprocal(std) * delta_mu, 1))


def _set_requires_grad(params, flag):
    for p in params:
        p.requires_grad = flag


def _trainable(model):
    """Names of parameters that are ever trainable (everything but the frozen word embedding, embedding.py:23)."""
    return [n for n, _ in model.named_parameters() if n != "embedding.word_embed.weight"]


def prep_disc_training(model):
    """cgn.py:363-373 (requires_grad toggles; .train()/.eval() only matters for dropout, which is explicit here)."""
    named = dict(model.named_parameters())
    for n in _trainable(model):
        named[n].requires_grad = _is_disc(n)


def _is_disc(n):
    return n.startswith("sentiment_discriminator.") or n.startswith("attribute_discriminators.")


def prep_enc_gen_training(model):
    """cgn.py:385-395."""
    named = dict(model.named_parameters())
    for n in _trainable(model):
        named[n].requires_grad = not _is_disc(n)


# --------------------------------------------------------------------------------------------------------------
# c inference — model/cgn.py:134-172
# --------------------------------------------------------------------------------------------------------------
def discriminator_forward(model, encoder_word_input, uniform):
    """cgn.py:151-172; t.bernoulli(softmax[:,1]) (:170) becomes (uniform < p1) with `uniform` ~ U[0,1)[B] given."""
    _scope("infer")
    with torch.no_grad():
        x = model.embedding.get_word_embed(encoder_word_input)  # :163
        logit = model.sentiment_discriminator(x)  # eval mode => no dropout (:151)
        p1 = F.softmax(logit, dim=1)[:, 1]  # :167
        if model.config.c_size == 1:
            return (uniform < p1).to(p1.dtype), p1  # :170-172
        # multi-attribute extension: one column of c per discriminator, uniform [B, C]
        p1s = [p1] + [F.softmax(d(x), dim=1)[:, 1] for d in model.attribute_discriminators]
        p = torch.stack(p1s, 1)
        return (uniform < p).to(p.dtype), p


# --------------------------------------------------------------------------------------------------------------
# discriminator training step — model/cgn.py:269-341
# --------------------------------------------------------------------------------------------------------------
def disc_train_step(model, optimizer, x_labelled, y_labelled, x_generated, c_generated, mask_labelled=None,
                    mask_generated=None, capture=None):
    """One step of the closure at cgn.py:269-341. x_*: int64 [B, L*] token batches (already padded,
    data_handler.py:87-102), y/c: class indices [B]. mask_*: explicit dropout masks (Sentiment_CNN is in .train()
    here, cgn.py:371). Returns the 5-tuple of cgn.py:341 (as tensors)."""
    cfg = model.config
    C = cfg.c_size
    total_loss = batch_train_ce_loss = generated_loss = generated_loss_ce = entropy = 0.0
    for a, D in enumerate(model.discriminators()):  # the reference: one discriminator (a = 0)
        y_l = y_labelled if C == 1 else y_labelled[:, a]
        c_g = c_generated.view(-1) if C == 1 else c_generated[:, a]
        _scope("disc_l%d" % a)
        logit = D(model.embedding.word_embed(x_labelled), mask_labelled)  # :303-304
        ce_l = F.cross_entropy(logit, y_l.long())  # :306
        _scope("disc_g%d" % a)
        logit_generated = D(model.embedding.word_embed(x_generated), mask_generated)  # :315-316
        log_softmax_generated = F.log_softmax(logit_generated, dim=1)  # :318
        softmax_generated = torch.exp(log_softmax_generated)  # :319
        ce_g = F.nll_loss(log_softmax_generated, c_g.long())  # :320
        ent = torch.neg(torch.mean(torch.sum(softmax_generated * log_softmax_generated, 1)))  # :327
        gen_l = ce_g + cfg.beta * ent  # :329
        total_loss = total_loss + ce_l + cfg.lambda_u * gen_l  # :334
        batch_train_ce_loss, generated_loss = batch_train_ce_loss + ce_l, generated_loss + gen_l
        generated_loss_ce, entropy = generated_loss_ce + ce_g, entropy + ent
    optimizer.zero_grad()  # :335
    total_loss.backward()  # :336
    if capture is not None:
        capture["disc_grads"] = {n: p.grad.detach().clone()
                                 for n, p in model.sentiment_discriminator.named_parameters() if p.grad is not None}
        capture["disc_grads"].update({"attribute_discriminators." + n: p.grad.detach().clone()
                                      for n, p in model.attribute_discriminators.named_parameters()
                                      if p.grad is not None})
    optimizer.step()  # :337
    return total_loss, batch_train_ce_loss, generated_loss, generated_loss_ce, entropy


# --------------------------------------------------------------------------------------------------------------
# encoder/generator training step — model/cgn.py:1049-1154
# --------------------------------------------------------------------------------------------------------------
def enc_gen_train_step(model, encoder_optimizer, generator_optimizer, batch, c_target, eps, global_step,
                       gen_drop_mask=None, calc_z_loss=True, eps_recon=None, kld_py2_floor=True, capture=None):
    """One step of the closure at cgn.py:1049-1154.

    batch = [encoder_word_input[B,L], encoder_character_input[B,L,W], gen_word_input[B,T], _, target[B,T]] (int64),
    c_target [B] float in {0,1}; eps [B,Z] the randn of encode_for_z (:96); eps_recon the second draw used by the
    z-reconstruction pass (:1127) when calc_z_loss. Returns the 8-tuple of :1154. `capture` (dict) receives the
    gradients seen by each optimizer.step() and `g` = encoder_grad_z_out at the time it is used (:1148)."""
    cfg = model.config
    encoder_optimizer.zero_grad()  # :1051
    generator_optimizer.zero_grad()  # :1052
    encoder_word_input, encoder_character_input, gen_word_input, _, target = batch
    batch_size = encoder_word_input.size(0)
    _scope("enc")
    encoder_input = model.embedding(encoder_word_input, encoder_character_input)  # :1063
    z_out, std_out, logvar_out, mu_out = model.encode_for_z(encoder_input, eps)  # :1065
    z = z_out.detach().clone().requires_grad_(True)  # :1067 Variable(z_out.data, requires_grad=True)
    c = c_target.detach().to(z.dtype)  # :1069
    kld_coeff = kld_coef(global_step, py2_floor=kld_py2_floor)  # :1074
    kld = (-0.5 * torch.sum(logvar_out - torch.pow(mu_out, 2) - torch.exp(logvar_out) + 1, 1)).mean().squeeze()  # :1076
    c = c.view(c.size(0), -1)  # :1078 (C columns; the reference has C = 1)
    input_code = torch.cat((z, c), 1)  # :1080
    generator_input = model.embedding.word_embed(gen_word_input)  # :1081
    logits, _ = model.generator(generator_input, input_code, gen_drop_mask, None)  # :1082
    logit_temperature = temp_coef(global_step)  # :1086
    logits = logits.view(-1, cfg.word_vocab_size) / logit_temperature  # :1087
    target = target.reshape(-1)  # :1088
    log_softmax = F.log_softmax(logits, dim=1)  # :1091
    cross_entropy = F.nll_loss(log_softmax, target)  # :1092 (mean over all B*T rows, pads included)
    total_cross_entropy = cross_entropy * 79  # :1096
    total_cross_entropy.backward(retain_graph=True)  # :1097
    encoder_grad_z_out = z.grad.data  # :1099 — an ALIAS of z.grad: the next backward accumulates into it in place
    softmax_output = torch.exp(log_softmax)  # :1103
    out_embedding = torch.mm(softmax_output, model.embedding.word_embed.weight)  # :1106
    out_embedding = out_embedding.view(batch_size, -1, out_embedding.size(1))[:, :-1, :]  # :1108
    _scope("soft0")
    c_reconstructed = model.sentiment_discriminator(out_embedding)  # :1114 (eval mode: no dropout)
    c_loss = F.cross_entropy(c_reconstructed, c.long()[:, 0])  # :1117-1118
    for a, D in enumerate(model.attribute_discriminators, 1):  # multi-attribute extension: one c-loss per attribute
        _scope("soft%d" % a)
        c_loss = c_loss + F.cross_entropy(D(out_embedding), c.long()[:, a])
    total_reconstruction_loss = cfg.lambda_c * c_loss  # :1120
    if calc_z_loss:  # :1122-1137
        _scope("recon")
        encoder_character_embedding = model.embedding.get_character_embed(encoder_character_input)
        encoder_reconstructed_input = torch.cat([out_embedding, encoder_character_embedding], dim=2)
        z_recon, std_recon, _, mu_recon = model.encode_for_z(encoder_reconstructed_input, eps_recon)
        mu = mu_out.detach()
        std = std_out.detach()
        mahalabonis_distance = compute_mahalabonis(mu, mu_recon, std, std_recon)
        z_loss = torch.sum(mahalabonis_distance) / batch_size
        total_reconstruction_loss = total_reconstruction_loss + cfg.lambda_z * z_loss
    else:
        z_loss = torch.zeros(1)  # :1139
    total_reconstruction_loss.backward()  # :1142
    if capture is not None:
        capture["gen_grads"] = {n: p.grad.detach().clone() for n, p in model.generator.named_parameters()}
        capture["g"] = encoder_grad_z_out.detach().clone()
        capture["z_grad_is_alias"] = bool(encoder_grad_z_out.data_ptr() == z.grad.data_ptr())
    generator_optimizer.step()  # :1143
    encoder_optimizer.zero_grad()  # :1145 (drops encoder grads produced by the z-reconstruction pass)
    total_kld_loss = kld_coeff * kld  # :1146
    total_kld_loss.backward(retain_graph=True)  # :1147
    z_out.backward(encoder_grad_z_out)  # :1148
    if capture is not None:
        enc_names = {id(p): n for n, p in model.named_parameters()}
        capture["enc_grads"] = {enc_names[id(p)]: p.grad.detach().clone() for p in model.encoder_params()
                                if p.grad is not None}
    encoder_optimizer.step()  # :1149
    total_encoder_loss = total_cross_entropy + total_kld_loss  # :1151
    total_generator_loss = total_cross_entropy + total_kld_loss + total_reconstruction_loss  # :1152
    return (cross_entropy, kld, z_loss, c_loss, kld_coeff, logit_temperature, total_encoder_loss,
            total_generator_loss)  # :1154


# --------------------------------------------------------------------------------------------------------------
# synthetic inputs — shared by tests, smoke() and bench.py so that every leg sees the same data
# --------------------------------------------------------------------------------------------------------------
def synthetic_word_embed(cfg, seed=1234):
    g = torch.Generator().manual_seed(seed)
    return ((torch.rand(cfg.word_vocab_size, cfg.word_embed_size, generator=g) * 2 - 1) * 0.25).numpy()


def synthetic_batch(cfg, B, L, seed=1234):
    """Fixed-length sentences laid out as utils/batch_loader.py:271-316 does: enc_* reversed, dec = [go]+s,
    target = s+[end]; the three special tokens are the last vocab indices (batch_loader.py:81-85)."""
    g = torch.Generator().manual_seed(seed)
    V = cfg.word_vocab_size
    s = torch.randint(0, V - 3, (B, L), generator=g)
    chars = torch.randint(0, cfg.char_vocab_size, (B, L, cfg.max_word_len), generator=g)
    go, end = V - 2, V - 1
    enc_word = torch.flip(s, dims=[1]).contiguous()
    enc_char = torch.flip(chars, dims=[1]).contiguous()
    dec_word = torch.cat([torch.full((B, 1), go, dtype=torch.int64), s], 1)
    target = torch.cat([s, torch.full((B, 1), end, dtype=torch.int64)], 1)
    out = dict(
        enc_word=enc_word, enc_char=enc_char, dec_word=dec_word, target=target,
        eps=torch.randn(B, cfg.latent_variable_size, generator=g),
        eps_recon=torch.randn(B, cfg.latent_variable_size, generator=g),
        c_uniform=torch.rand(B, generator=g),
        x_labelled=torch.randint(0, V - 3, (B, L), generator=g),
        y_labelled=torch.randint(0, 2, (B,), generator=g),
        x_generated=torch.randint(0, V - 3, (B, L), generator=g),
        c_generated=torch.randint(0, 2, (B,), generator=g),
    )
    C = getattr(cfg, "c_size", 1)
    if C > 1:  # multi-attribute extension: one label / code / uniform column per attribute (column 0 as above)
        for k, draw in (("c_uniform", lambda: torch.rand(B, generator=g)),
                        ("y_labelled", lambda: torch.randint(0, 2, (B,), generator=g)),
                        ("c_generated", lambda: torch.randint(0, 2, (B,), generator=g))):
            out[k] = torch.stack([out[k]] + [draw() for _ in range(C - 1)], 1)
    return out


def wake_sleep_step(model, opts, data, global_step, calc_z_loss=False, masks=None, capture=None):
    """train_functions.py:49-108 with the sampler replaced by the synthetic 'generated' batch:
    disc step -> c inference -> enc/gen step. opts = (disc_opt, enc_opt, gen_opt)."""
    masks = masks or {}
    disc_opt, enc_opt, gen_opt = opts
    prep_disc_training(model)  # :55
    d_losses = disc_train_step(model, disc_opt, data["x_labelled"], data["y_labelled"], data["x_generated"],
                               data["c_generated"], masks.get("labelled"), masks.get("generated"), capture)  # :57
    _set_requires_grad([p for d in model.discriminators() for p in d.parameters()], False)  # :81
    c_batch, _ = discriminator_forward(model, data["enc_word"], data["c_uniform"])  # :82
    prep_enc_gen_training(model)  # :85
    batch = [data["enc_word"], data["enc_char"], data["dec_word"], None, data["target"]]
    eg_losses = enc_gen_train_step(model, enc_opt, gen_opt, batch, c_batch, data["eps"], global_step,
                                   masks.get("gen_drop"), calc_z_loss, data["eps_recon"], capture=capture)  # :88-90
    return d_losses, c_batch, eg_losses


def make_optimizers(model):
    """cgn.py:267 and :1046-1047 (torch Adam defaults, lr = config.learning_rate)."""
    lr = model.config.learning_rate
    prep_disc_training(model)
    disc = torch.optim.Adam(model.sentiment_discriminator_parameters(), lr)
    prep_enc_gen_training(model)
    enc = torch.optim.Adam(model.encoder_params(), lr)
    gen = torch.optim.Adam(model.generator_params(), lr)
    return disc, enc, gen


# --------------------------------------------------------------------------------------------------------------
# Sleep-phase beam sampler: model/cgn.py:747-921 (sample_from_generator), utils/beam_search.py:16-153 (Beam),
# model/generator.py:24-43 (only_decoder_beam). Restated per sentence: in the reference the sentences advance in
# lock-step but never interact, so one sentence = one Beam + its own [layers, beam, H] LSTM state.
# Conscious deviation: the reference tiles the seed `zc` over ALL sentences even after finished ones
# were compacted away (generator.py:30-32 vs cgn.py:845-869), which pairs the remaining sentences with the wrong
# seeds; here every sentence keeps its own seed. Until the first sentence finishes the two are identical.
# Python-2 integer divisions (beam_search.py:88 `bestScoresId / num_words`) are written as floor divisions.
# --------------------------------------------------------------------------------------------------------------
def sample_from_generator(model, seed, seq_len, beam_size, go, eos, pad, n_best=1, learning=False):
    """seed [S, Z+C]. Returns (hyps, scores): hyps[s] = list over n_best of token lists (the walk-back of
    Beam.get_hyp, beam_search.py:121-132), scores[s] = the beam's final scores sorted descending (sort_best :101).
    learning=True (cgn.py:880,901-902): returns instead the list over sentences of [len_s, V] log_softmax rows collected
    by Beam.get_hyp_probs (beam_search.py:140-152)."""
    gen, emb = model.generator, model.embedding.word_embed
    V = gen.fc.out_features
    hyps, all_scores, all_probs = [], [], []
    with torch.no_grad():
        for s in range(seed.shape[0]):
            output_probs = []  # beam_search.py:43
            zc = seed[s:s + 1]
            scores = torch.zeros(beam_size, dtype=seed.dtype)  # beam_search.py:31
            prev_ks, next_ys = [], [torch.full((beam_size,), pad, dtype=torch.int64)]  # :34-37
            next_ys[0][0] = go  # :38
            state = None
            for _ in range(seq_len):  # cgn.py:786
                x = emb(next_ys[-1]).unsqueeze(1)  # cgn.py:793  [K,1,E]
                x = torch.cat([x, zc.expand(beam_size, -1).unsqueeze(1)], 2)  # generator.py:30-39
                out, state = gen.rnn(x, state)  # generator.py:41
                word_lk = F.log_softmax(gen.fc(out.squeeze(1)), dim=1)  # cgn.py:800-803  [K,V]
                output_probs.append(word_lk)  # beam_search.py:73
                beam_lk = word_lk + scores.unsqueeze(1) if prev_ks else word_lk[0]  # beam_search.py:76-79
                best, best_id = beam_lk.reshape(-1).topk(beam_size, 0, True, True)  # :83
                scores = best
                prev_k = best_id // V  # :88
                prev_ks.append(prev_k)
                next_ys.append(best_id - prev_k * V)  # :92
                state = tuple(st.index_select(1, prev_k) for st in state)  # cgn.py:823-838
                if next_ys[-1][0].item() == eos:  # beam_search.py:96-97
                    break
            sc, ks = torch.sort(scores, 0, True)  # :101-103
            out_h = []
            for k in ks[:n_best].tolist():
                hyp = []
                for j in range(len(prev_ks) - 1, -1, -1):  # :127-129
                    hyp.append(int(next_ys[j + 1][k]))
                    k = int(prev_ks[j][k])
                out_h.append(hyp[::-1])
            hyps.append(out_h)
            all_scores.append(sc)
            k, rows = 0, []
            for j in range(len(prev_ks) - 1, -1, -1):  # get_hyp_probs, beam_search.py:140-152
                k = int(prev_ks[j][k])  # :148 (the parent FIRST, then its row: :149)
                rows.append(output_probs[j][k])
            all_probs.append(torch.stack(rows[::-1]))
    if learning:
        return all_probs
    return hyps, all_scores
