"""Generator: [word embedding ‖ tiled (z,c)] -> 2-layer LSTM -> vocabulary projection.
Same class, attributes (rnn, fc) and signatures as the reference's model/generator.py:10-64."""
import torch as t
import torch.nn as nn

from ..hip import nn as hnn
from ..hip import ops


class _GenInputFn(t.autograd.Function):
    """cat(generator_input * mask, tile(zc over time), dim=2)  (generator.py:52-56) with HIP copy kernels."""

    @staticmethod
    def forward(ctx, gen_in, zc, mask):
        hnn.need_cuda(gen_in, "Generator")
        B, T, E = gen_in.shape
        Zc = zc.shape[1]
        x = hnn.new((B, T, E + Zc), gen_in.device)
        g = hnn.f32c(gen_in)
        if mask is not None:
            m = hnn.f32c(mask).view(B * T, E)
            gm = hnn.new((B * T, E), gen_in.device)
            # elementwise product via the row-dot-free path: copy then scale by mask is a gather with identity rows;
            # use copy2d twice: x = g ; x *= mask  -> done with the mask-aware gather kernel's sibling below
            ops.mul2d(g.view(B * T, E), m, gm)
            g = gm.view(B, T, E)
            ctx.mask = m
        else:
            ctx.mask = None
        ops.copy3d(g, x[:, :, :E])
        ops.copy3d(hnn.f32c(zc).unsqueeze(1).expand(B, T, Zc), x[:, :, E:])
        ctx.dims = (B, T, E, Zc)
        return x

    @staticmethod
    def backward(ctx, dx):
        B, T, E, Zc = ctx.dims
        d = hnn.f32c(dx)
        dg = dz = None
        if ctx.needs_input_grad[0]:
            dg = hnn.f32c(d[:, :, :E])
            if ctx.mask is not None:
                out = hnn.new((B * T, E), d.device)
                ops.mul2d(dg.view(B * T, E), ctx.mask, out)
                dg = out.view(B, T, E)
        if ctx.needs_input_grad[1]:
            dzt = hnn.f32c(d[:, :, E:])
            dz = hnn.new((B, Zc), d.device)
            ops.sum_time(dzt, dz)
        return dg, dz, None


def _dropout_mask(x, drop_prob, training, legacy_noop):
    """F.dropout(x, drop_prob) of generator.py:52 / decoder.py:64.

    In the torch versions the reference targets (<= 0.3) the functional default is training=False, i.e. the call is a
    no-op; `legacy_noop=True` (default) keeps that. With legacy_noop=False the mask is drawn (torch RNG, Bernoulli
    keep-probability 1-p, scaled 1/(1-p)) whenever the module is in training mode. A tensor `drop_prob` is taken as
    the explicit, already scaled mask (used by the parity tests)."""
    if isinstance(drop_prob, t.Tensor):
        return drop_prob
    if legacy_noop or not training or not drop_prob:
        return None
    keep = 1.0 - float(drop_prob)
    return (t.rand(x.shape, device=x.device) < keep).to(t.float32) / keep


class Generator(nn.Module):
    """Attributes `rnn` (LSTM over [word embedding | z | c]) and `fc` (hidden -> vocabulary), as in the reference;
    the width of c is config.c_size (the reference hard-codes 1, generator.py:17,54)."""

    legacy_dropout_noop = True

    def __init__(self, params):
        super().__init__()
        self.params = params
        code = params.latent_variable_size + getattr(params, "c_size", 1)
        H = params.decoder_rnn_size
        self.rnn = hnn.LSTM(input_size=code + params.word_embed_size, hidden_size=H,
                            num_layers=params.decoder_num_layers, batch_first=True)
        self.fc = hnn.Linear(H, params.word_vocab_size)

    def _run_rnn(self, words, code_rows, drop_prob, state):
        """LSTM over words [rows, steps, E] with one code row [rows, Z+C] repeated at every step."""
        mask = _dropout_mask(words, drop_prob, self.training, self.legacy_dropout_noop)
        return self.rnn(_GenInputFn.apply(words, code_rows, mask), state)

    def only_decoder_beam(self, generator_input, zc, drop_prob, initial_state=None):
        """generator.py:24-43 (sampler): the LSTM stack alone, state carried between calls. The zc block is repeated
        whole until it covers the beam batch (row i gets seed i mod S, integer division as under Python 2)."""
        rows = generator_input.size(0)
        S = zc.size(0)
        code_rows = zc.repeat(rows // S, 1)
        return self._run_rnn(generator_input, code_rows, drop_prob, initial_state)

    def forward(self, generator_input, zc, drop_prob, initial_state=None):
        """generator.py:45-64: logits [B, T, V] and the LSTM's final (h_n, c_n)."""
        B, T = generator_input.shape[:2]
        hidden, final_state = self._run_rnn(generator_input, zc, drop_prob, initial_state)
        logits = self.fc(hidden.reshape(B * T, -1))
        return logits.view(B, T, -1), final_state
