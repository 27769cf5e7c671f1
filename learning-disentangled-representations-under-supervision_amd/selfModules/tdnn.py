"""Character-level word encoder (char-CNN / "TDNN") on MI355X: for every kernel width a bias-free 1-D convolution over
the word's character embeddings, tanh, max over positions; the per-width features are concatenated.

Drop-in for the reference's selfModules/tdnn.py:8-52: class `TDNN(params)`, kernels `[out_dim, char_embed_size, kW]`
drawn from U(-1, 1) for each `(kW, out_dim)` of `params.kernels` and registered as `TDNN_kernel-{i}` (checkpoint
names), `forward(x[B, L, W, Cc]) -> [B, L, sum(out_dim)]`. All widths run as ONE grouped MFMA launch plus one pooling
launch (hip/engine.py `tdnn_conv_fwd`; backward `tdnn_conv_bwd`), wrapped for autograd by hip/nn.py `_TDNNFn`.
"""
import torch
import torch.nn as nn

from ..hip import nn as hnn


class TDNN(nn.Module):
    def __init__(self, params):
        super().__init__()
        self.params = params
        self.kernels = []
        for i, (width, maps) in enumerate(params.kernels):
            kernel = nn.Parameter(torch.empty(maps, params.char_embed_size, width).uniform_(-1, 1))
            self.register_parameter('TDNN_kernel-%d' % i, kernel)
            self.kernels.append(kernel)

    def forward(self, x):
        """x: [batch, seq_len, max_word_len, char_embed_size] character embeddings -> [batch, seq_len, sum_depth]."""
        if x.dim() != 4:
            raise AssertionError('Wrong input rang, must be equal to 4, but {} found'.format(x.dim()))
        B, L, W, Cc = x.shape
        want = (self.params.max_word_len, self.params.char_embed_size)
        if (W, Cc) != want:
            raise AssertionError('Wrong word length / embedding size, must be equal to {}, but {} found'.format(
                want, (W, Cc)))
        words = x.reshape(B * L, W, Cc)  # one "sentence" of W characters per word
        return hnn._TDNNFn.apply(words, *self.kernels).view(B, L, -1)
