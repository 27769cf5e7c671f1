"""Legacy unconditioned decoder (the Generator without the structured code c). Same class and signatures as the
reference's model/decoder.py:8-76; the reference's CGN never instantiates it (cgn.py:13-17), it shares the Generator's
kernels."""
import torch.nn as nn

from ..hip import nn as hnn
from ..utils.functional import parameters_allocation_check
from .generator import _GenInputFn, _dropout_mask


class Decoder(nn.Module):

    legacy_dropout_noop = True

    def __init__(self, params):
        super(Decoder, self).__init__()

        self.params = params

        self.rnn = hnn.LSTM(input_size=self.params.latent_variable_size + self.params.word_embed_size,
                            hidden_size=self.params.decoder_rnn_size,
                            num_layers=self.params.decoder_num_layers,
                            batch_first=True)

        self.fc = hnn.Linear(self.params.decoder_rnn_size, self.params.word_vocab_size)

    def only_decoder_beam(self, decoder_input, z, drop_prob, initial_state=None):
        assert parameters_allocation_check(self), \
            'Invalid CUDA options. Parameters should be allocated in the same memory'
        [beam_batch_size, _, _] = decoder_input.size()
        mask = _dropout_mask(decoder_input, drop_prob, self.training, self.legacy_dropout_noop)
        # decoder.py:34-35: one latent vector shared by every beam
        z = z.reshape(1, -1).expand(beam_batch_size, -1)
        x = _GenInputFn.apply(decoder_input, z, mask)
        rnn_out, final_state = self.rnn(x, initial_state)
        return rnn_out, final_state

    def forward(self, decoder_input, z, drop_prob, initial_state=None):
        """
        :param decoder_input: [batch_size, seq_len, embed_size]
        :param z: [batch_size, latent_variable_size]
        :return: logits [batch_size, seq_len, word_vocab_size], final rnn state
        """
        assert parameters_allocation_check(self), \
            'Invalid CUDA options. Parameters should be allocated in the same memory'

        [batch_size, seq_len, _] = decoder_input.size()
        mask = _dropout_mask(decoder_input, drop_prob, self.training, self.legacy_dropout_noop)
        x = _GenInputFn.apply(decoder_input, z, mask)

        rnn_out, final_state = self.rnn(x, initial_state)
        rnn_out = rnn_out.contiguous().view(-1, self.params.decoder_rnn_size)

        result = self.fc(rnn_out)
        result = result.view(batch_size, seq_len, self.params.word_vocab_size)

        return result, final_state
