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

    legacy_dropout_noop = True

    def __init__(self, params):
        super(Generator, self).__init__()

        self.params = params
        c_size = getattr(params, "c_size", 1)

        self.rnn = hnn.LSTM(input_size=(self.params.latent_variable_size + c_size) + self.params.word_embed_size,
                            hidden_size=self.params.decoder_rnn_size,
                            num_layers=self.params.decoder_num_layers,
                            batch_first=True)

        self.fc = hnn.Linear(self.params.decoder_rnn_size, self.params.word_vocab_size)

    def only_decoder_beam(self, generator_input, zc, drop_prob, initial_state=None):
        """Single call of the LSTM stack with carried state, generator.py:24-43 (used by the sampler)."""
        [beam_batch_size, steps, _] = generator_input.size()
        mask = _dropout_mask(generator_input, drop_prob, self.training, self.legacy_dropout_noop)
        # generator.py:30-32: every seed's zc is repeated for its beams (Python-2 integer division in the reference)
        reps = beam_batch_size // zc.size(0)
        zc = zc.unsqueeze(0).expand(reps, zc.size(0), zc.size(1)).reshape(beam_batch_size, -1)
        x = _GenInputFn.apply(generator_input, zc, mask)
        rnn_out, final_state = self.rnn(x, initial_state)
        return rnn_out, final_state

    def forward(self, generator_input, zc, drop_prob, initial_state=None):
        [batch_size, seq_len, _] = generator_input.size()

        mask = _dropout_mask(generator_input, drop_prob, self.training, self.legacy_dropout_noop)
        x = _GenInputFn.apply(generator_input, zc, mask)

        rnn_out, final_state = self.rnn(x, initial_state)
        rnn_out = rnn_out.contiguous().view(-1, self.params.decoder_rnn_size)

        result = self.fc(rnn_out)
        result = result.view(batch_size, seq_len, self.params.word_vocab_size)

        return result, final_state
