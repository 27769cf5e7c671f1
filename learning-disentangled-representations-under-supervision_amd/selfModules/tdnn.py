"""Char-CNN word encoder. Same class/parameter names as the reference's selfModules/tdnn.py:8-52; the convolutions run
as implicit-GEMM MFMA kernels with a fused tanh + max-over-positions (hip/engine.py: tdnn_conv_fwd / _bwd)."""
import torch as t
import torch.nn as nn
from torch.nn import Parameter

from ..hip import nn as hnn


class TDNN(nn.Module):
    def __init__(self, params):
        super(TDNN, self).__init__()

        self.params = params

        # tdnn.py:14-16: bias-free kernels [out_dim, char_embed_size, kW], U(-1, 1), registered as TDNN_kernel-{i}
        self.kernels = [Parameter(t.Tensor(out_dim, self.params.char_embed_size, kW).uniform_(-1, 1))
                        for kW, out_dim in params.kernels]
        self._add_to_parameters(self.kernels, 'TDNN_kernel')

    def forward(self, x):
        """
        :param x: tensor with shape [batch_size, max_seq_len, max_word_len, char_embed_size]
        :return: tensor with shape [batch_size, max_seq_len, depth_sum]
        """
        input_size = x.size()
        assert len(input_size) == 4, \
            'Wrong input rang, must be equal to 4, but {} found'.format(len(input_size))

        [batch_size, seq_len, word_len, embed_size] = input_size

        assert embed_size == self.params.char_embed_size, \
            'Wrong embedding size, must be equal to {}, but {} found'.format(self.params.char_embed_size, embed_size)
        assert word_len == self.params.max_word_len, \
            'Wrong word length, must be equal to {}, but {} found'.format(self.params.max_word_len, word_len)

        y = hnn._TDNNFn.apply(x.reshape(batch_size * seq_len, word_len, embed_size), *self.kernels)
        return y.view(batch_size, seq_len, -1)

    def _add_to_parameters(self, parameters, name):
        for i, parameter in enumerate(parameters):
            self.register_parameter(name='{}-{}'.format(name, i), param=parameter)
