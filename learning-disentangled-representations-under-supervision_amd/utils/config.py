"""Hyper-parameters. Same constructor and attribute names as the reference's utils/config.py:4-37; the dimensions the
reference hard-codes are keyword arguments here (defaults = the reference's values) because BASELINE.json's
configurations use other sizes (H 256/512/1024, Z 32/64, Ks 1..5, Co 128)."""


class Config:
    def __init__(self, max_word_len, max_seq_len, word_vocab_size, char_vocab_size, learning_rate, lambda_c, lambda_z,
                 lambda_u, beta, *, word_embed_size=300, char_embed_size=15,
                 kernels=((1, 25), (2, 50), (3, 75), (4, 100), (5, 125), (6, 150)),
                 encoder_rnn_size=600, encoder_num_layers=1, latent_variable_size=1100,
                 decoder_rnn_size=800, decoder_num_layers=2,
                 sentiment_kernel_size=(3, 4, 5), sentiment_kernel_num=100, sentiment_dropout=0.3, c_size=1):
        self.learning_rate = learning_rate
        self.max_word_len = int(max_word_len)
        self.max_seq_len = int(max_seq_len) + 1  # go or eos token (config.py:9)

        self.word_vocab_size = int(word_vocab_size)
        self.char_vocab_size = int(char_vocab_size)

        self.word_embed_size = int(word_embed_size)
        self.char_embed_size = int(char_embed_size)

        self.kernels = [tuple(k) for k in kernels]
        self.sum_depth = sum(depth for _, depth in self.kernels)

        self.encoder_rnn_size = int(encoder_rnn_size)
        self.encoder_num_layers = int(encoder_num_layers)
        self.latent_variable_size = int(latent_variable_size)
        self.decoder_rnn_size = int(decoder_rnn_size)
        self.decoder_num_layers = int(decoder_num_layers)

        # Sentiment discriminator
        self.sentiment_kernel_size = list(sentiment_kernel_size)
        self.sentiment_kernel_num = int(sentiment_kernel_num)
        self.sentiment_dropout = sentiment_dropout

        # Loss weights
        self.lambda_c = lambda_c
        self.lambda_u = lambda_u
        self.lambda_z = lambda_z
        self.beta = beta

        # width of the structured code c; the reference hard-codes 1 (model/generator.py:17,54)
        self.c_size = int(c_size)

        if self.encoder_num_layers != 1:
            raise NotImplementedError("the HIP encoder path implements the reference's 1-layer biLSTM (config.py:21)")
        if self.encoder_rnn_size % 4 or self.decoder_rnn_size % 4:
            raise ValueError("LSTM hidden sizes must be multiples of 4 (float4 operand loads)")
        if self.max_word_len < max(k for k, _ in self.kernels):
            raise ValueError("max_word_len must be >= the widest TDNN kernel (tdnn.py:42 would raise)")
