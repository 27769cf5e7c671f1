"""Word + character embedding front end. Same class/attribute names as the reference's selfModules/embedding.py:10-74:
frozen word table loaded from a .npy file, trainable char table (U(-1,1)), TDNN over characters, concat on dim 2."""
import numpy as np
import torch as t
import torch.nn as nn
from torch.nn import Parameter

from ..hip import nn as hnn
from .tdnn import TDNN


class _Table(nn.Module):
    """nn.Embedding stand-in: a `.weight` parameter and a HIP gather forward."""

    def __init__(self, weight, requires_grad):
        super(_Table, self).__init__()
        self.weight = Parameter(weight, requires_grad=requires_grad)

    def forward(self, idx):
        return hnn.embedding_lookup(idx, self.weight)


class Embedding(nn.Module):
    def __init__(self, params, emb_file):
        """emb_file: path of the word-embedding .npy (embedding.py:16); an ndarray is accepted too."""
        super(Embedding, self).__init__()

        self.params = params

        word_embed = np.load(emb_file) if isinstance(emb_file, str) else np.asarray(emb_file)
        assert word_embed.shape == (self.params.word_vocab_size, self.params.word_embed_size), \
            'word embedding file has shape {}, config says {}'.format(
                word_embed.shape, (self.params.word_vocab_size, self.params.word_embed_size))

        # embedding.py:23: the word table is a constant (requires_grad=False)
        self.word_embed = _Table(t.from_numpy(word_embed).float().clone(), requires_grad=False)
        # embedding.py:24-25
        self.char_embed = _Table(
            t.Tensor(self.params.char_vocab_size, self.params.char_embed_size).uniform_(-1, 1), requires_grad=True)

        self.TDNN = TDNN(self.params)

    def forward(self, word_input, character_input):
        """word_input [B, L] int64, character_input [B, L, W] int64 -> [B, L, word_embed_size + sum_depth]
        (word vectors | char-TDNN features), embedding.py:29-55."""
        if tuple(word_input.shape[:2]) != tuple(character_input.shape[:2]):
            raise AssertionError('Word input and character input must have the same sizes, but {} and {} found'.format(
                tuple(word_input.shape), tuple(character_input.shape)))
        return _Cat2Fn.apply(self.get_word_embed(word_input), self.get_character_embed(character_input))

    def get_word_embed(self, word_input):
        return self.word_embed(word_input)

    def get_character_embed(self, character_input):
        """[B, L, W] int64 -> [B, L, sum_depth]: char table lookup, then the TDNN (embedding.py:60-74)."""
        B, L, W = character_input.shape
        p = self.params
        chars = self.char_embed(character_input.reshape(B * L, W))
        return self.TDNN(chars.view(B, L, p.max_word_len, p.char_embed_size))


class _Cat2Fn(t.autograd.Function):
    """t.cat([a, b], 2) for [B,L,*] tensors with HIP strided copies (embedding.py:53)."""

    @staticmethod
    def forward(ctx, a, b):
        from ..hip import ops
        B, L, Ea = a.shape
        Eb = b.shape[2]
        out = hnn.new((B, L, Ea + Eb), a.device)
        o2 = out.view(B * L, Ea + Eb)
        ops.copy2d(hnn.f32c(a).view(B * L, Ea), o2[:, :Ea])
        ops.copy2d(hnn.f32c(b).view(B * L, Eb), o2[:, Ea:])
        ctx.split = (Ea, Eb)
        return out

    @staticmethod
    def backward(ctx, dout):
        Ea, Eb = ctx.split
        d = hnn.f32c(dout)
        da = hnn.f32c(d[:, :, :Ea]) if ctx.needs_input_grad[0] else None
        db = hnn.f32c(d[:, :, Ea:]) if ctx.needs_input_grad[1] else None
        return da, db
