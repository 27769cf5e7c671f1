"""Sentence encoder on MI355X: highway stack over every token row, then one bidirectional LSTM layer; the context
handed to e2mu / e2logvar is the concatenation of the two directions' final CELL states.

Drop-in for the reference's model/encoder.py:9-48: class `Encoder(params)`, sub-modules `hw1` and `rnn` (hence the
state_dict keys `hw1.*`, `rnn.weight_ih_l0[_reverse]` ...), `forward(input, State) -> (context, h_n, c_n)`.
Note on the reference (encoder.py:40-46): it unpacks nn.LSTM's `(h_n, c_n)` as `(transfer_state_1, final_state)`, so
its "final_state" — and therefore the context — is c_n, not h_n. That is reproduced here on purpose.
"""
import torch as t
import torch.nn as nn
import torch.nn.functional as F

from ..hip import nn as hnn
from ..selfModules.highway import Highway


class _ConcatColumns(t.autograd.Function):
    """[a | b] along dim 1 for two [B, H] tensors, as two strided HIP copies; backward splits the columns again."""

    @staticmethod
    def forward(ctx, a, b):
        from ..hip import ops
        ha, hb = a.shape[1], b.shape[1]
        joined = hnn.new((a.shape[0], ha + hb), a.device)
        ops.copy2d(hnn.f32c(a), joined[:, :ha])
        ops.copy2d(hnn.f32c(b), joined[:, ha:])
        ctx.split = ha
        return joined

    @staticmethod
    def backward(ctx, grad):
        g = hnn.f32c(grad)
        return hnn.f32c(g[:, :ctx.split]), hnn.f32c(g[:, ctx.split:])


class Encoder(nn.Module):
    def __init__(self, params):
        super().__init__()
        self.params = params
        width = params.word_embed_size + params.sum_depth  # [word embedding | char-TDNN features]
        self.hw1 = Highway(width, 2, F.relu)
        self.rnn = hnn.LSTM(input_size=width, hidden_size=params.encoder_rnn_size,
                            num_layers=params.encoder_num_layers, batch_first=True, bidirectional=True)

    def forward(self, input, State=None):
        """input [B, L, E+S]; State: optional (h0, c0) for the LSTM.
        Returns (context [B, 2*He] = [c_n forward | c_n backward] of the last layer, h_n, c_n)."""
        if not all(p.is_cuda == input.is_cuda for p in self.parameters()):
            # the reference asserts this through parameters_allocation_check (encoder.py:35-36)
            raise AssertionError('Invalid CUDA options. Parameters should be allocated in the same memory')
        B, L, D = input.shape
        tokens = self.hw1(input.reshape(B * L, D)).view(B, L, D)
        _, (h_n, c_n) = self.rnn(tokens, State)
        He, layers = self.params.encoder_rnn_size, self.params.encoder_num_layers
        last = c_n.view(layers, 2, B, He)[layers - 1]
        return _ConcatColumns.apply(last[0], last[1]), h_n, c_n
