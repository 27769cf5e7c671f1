"""Sentence encoder: Highway(Din, 2, relu) -> 1-layer bidirectional LSTM; returns the concatenated final CELL states.
Same class, attribute names (hw1, rnn) and return tuple as the reference's model/encoder.py:9-48."""
import torch as t
import torch.nn as nn
import torch.nn.functional as F

from ..hip import nn as hnn
from ..selfModules.highway import Highway
from ..utils.functional import parameters_allocation_check


class Encoder(nn.Module):
    def __init__(self, params):
        super(Encoder, self).__init__()

        self.params = params

        self.hw1 = Highway(self.params.sum_depth + self.params.word_embed_size, 2, F.relu)

        self.rnn = hnn.LSTM(input_size=self.params.word_embed_size + self.params.sum_depth,
                            hidden_size=self.params.encoder_rnn_size,
                            num_layers=self.params.encoder_num_layers,
                            batch_first=True,
                            bidirectional=True)

    def forward(self, input, State=None):
        """
        :param input: [batch_size, seq_len, embed_size] tensor
        :return: (context [batch_size, 2*encoder_rnn_size], h_n, c_n)
        """
        [batch_size, seq_len, embed_size] = input.size()

        input = input.reshape(-1, embed_size)
        input = self.hw1(input)
        input = input.view(batch_size, seq_len, embed_size)

        assert parameters_allocation_check(self), \
            'Invalid CUDA options. Parameters should be allocated in the same memory'

        # encoder.py:40: the reference unpacks (h_n, c_n) as (transfer_state_1, final_state): "final_state" is c_n
        _, (transfer_state_1, final_state) = self.rnn(input, State)
        transfer_state_2 = final_state

        final_state = final_state.view(self.params.encoder_num_layers, 2, batch_size, self.params.encoder_rnn_size)
        final_state = final_state[-1]
        context = _Cat1Fn.apply(final_state[0], final_state[1])

        return context, transfer_state_1, transfer_state_2


class _Cat1Fn(t.autograd.Function):
    """t.cat([a, b], 1) for [B,H] tensors (encoder.py:46) with HIP copies."""

    @staticmethod
    def forward(ctx, a, b):
        from ..hip import ops
        out = hnn.new((a.shape[0], a.shape[1] + b.shape[1]), a.device)
        ops.copy2d(hnn.f32c(a), out[:, :a.shape[1]])
        ops.copy2d(hnn.f32c(b), out[:, a.shape[1]:])
        ctx.h = a.shape[1]
        return out

    @staticmethod
    def backward(ctx, dout):
        d = hnn.f32c(dout)
        return hnn.f32c(d[:, :ctx.h]), hnn.f32c(d[:, ctx.h:])
