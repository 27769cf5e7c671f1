"""Legacy unconditioned decoder: the Generator without the structured code c (input = [word embedding | z]).

Drop-in for the reference's model/decoder.py:8-76 — class `Decoder(params)`, sub-modules `rnn` / `fc`,
`forward(decoder_input, z, drop_prob, initial_state)` and `only_decoder_beam(...)`. The reference's CGN never
instantiates it (cgn.py:13-17 import only the Generator); it is kept because the north star names the file, and it
runs on exactly the Generator's kernels (hip/nn.py LSTM / Linear, `_GenInputFn` for the tiled-z concat).
"""
import torch.nn as nn

from ..hip import nn as hnn
from .generator import _GenInputFn, _dropout_mask


class Decoder(nn.Module):

    legacy_dropout_noop = True  # F.dropout(x, p) of decoder.py:64 is a no-op in the torch versions it targets

    def __init__(self, params):
        super().__init__()
        self.params = params
        p = params
        self.rnn = hnn.LSTM(input_size=p.word_embed_size + p.latent_variable_size, hidden_size=p.decoder_rnn_size,
                            num_layers=p.decoder_num_layers, batch_first=True)
        self.fc = hnn.Linear(p.decoder_rnn_size, p.word_vocab_size)

    def _check_placement(self, x):
        if not all(q.is_cuda == x.is_cuda for q in self.parameters()):
            raise AssertionError('Invalid CUDA options. Parameters should be allocated in the same memory')

    def _rnn_input(self, decoder_input, z, drop_prob):
        mask = _dropout_mask(decoder_input, drop_prob, self.training, self.legacy_dropout_noop)
        return _GenInputFn.apply(decoder_input, z, mask)

    def only_decoder_beam(self, decoder_input, z, drop_prob, initial_state=None):
        """One LSTM call with carried state for the sampler; a single latent vector is shared by every beam row
        (decoder.py:34-35). Returns (rnn_out, final_state)."""
        self._check_placement(decoder_input)
        rows = decoder_input.size(0)
        return self.rnn(self._rnn_input(decoder_input, z.reshape(1, -1).expand(rows, -1), drop_prob), initial_state)

    def forward(self, decoder_input, z, drop_prob, initial_state=None):
        """decoder_input [B, T, E], z [B, Z] -> (logits [B, T, V], final (h_n, c_n))."""
        self._check_placement(decoder_input)
        B, T, _ = decoder_input.shape
        hidden, final_state = self.rnn(self._rnn_input(decoder_input, z, drop_prob), initial_state)
        logits = self.fc(hidden.reshape(B * T, self.params.decoder_rnn_size))
        return logits.view(B, T, self.params.word_vocab_size), final_state
