"""DataHandler with the interface of the reference's utils/data_handler.py for corpora held in memory: the generator
corpus (a BatchLoader) plus the labelled sentiment sentences, split 2:1 into train / dev (data_handler.py:36-67).
Same method names and return layouts as the reference for what the step closures and the training loop touch
(get_sentiment_train_batch, get_sentiment_dev_batch, create_sentiment_batch, num_*_sentiment_batches, batch_size,
gen_batch_loader); create_sentiment_batch is pinned against the reference's own (tests/golden/reference_batches.npz).
Not rebuilt: the pickled-file loading with its hard-coded paths (data_handler.py:19-34)."""
import numpy as np
import torch as t

from .batch_loader import BatchLoader


class DataHandler:
    def __init__(self, gen_sentences, sentiment_X, sentiment_Y, batch_size, train_lines=None, seed=0,
                 batch_loader=None):
        """gen_sentences: corpus for the encoder/generator; sentiment_X: labelled sentences (strings); sentiment_Y:
        their labels (0/1). Words of sentiment_X outside the generator vocabulary map to the pad token (the reference
        would produce None there, data_handler.py:65)."""
        self.batch_size = int(batch_size)
        self.gen_batch_loader = batch_loader or BatchLoader(gen_sentences, train_lines)
        self.vocab_size = self.gen_batch_loader.words_vocab_size
        n = len(sentiment_X)
        rng = np.random.RandomState(seed)
        indices = rng.permutation(n)  # data_handler.py:52-55: shuffled once, 2:1 split
        self.train_sentiment_size = (2 * n) // 3
        self.dev_sentiment_size = n - self.train_sentiment_size
        self.num_train_sentiment_batches = self.train_sentiment_size // self.batch_size
        self.num_dev_sentiment_batches = self.dev_sentiment_size // self.batch_size
        self.sentiment_train_indices = indices[:self.train_sentiment_size]
        self.sentiment_dev_indices = indices[self.train_sentiment_size:]
        w2i = self.gen_batch_loader.word_to_idx
        pad = w2i[self.gen_batch_loader.pad_token]
        self.sentiment_discriminator_X = [[w2i.get(w, pad) for w in line.split()] for line in sentiment_X]
        self.sentiment_discriminator_Y = [int(y) for y in sentiment_Y]

    # data_handler.py:69-85
    def _batch(self, which, batch_index):
        ids = np.arange(batch_index * self.batch_size, (batch_index + 1) * self.batch_size)
        X = [list(self.sentiment_discriminator_X[which[x]]) for x in ids]
        Y = [self.sentiment_discriminator_Y[which[x]] for x in ids]
        return X, Y

    def get_sentiment_train_batch(self, batch_index):
        return self._batch(self.sentiment_train_indices, batch_index)

    def get_sentiment_dev_batch(self, batch_index):
        return self._batch(self.sentiment_dev_indices, batch_index)

    # data_handler.py:87-102
    def create_sentiment_batch(self, x_dg):
        """Pads every sentence to the longest of the batch with the pad token; returns an int64 tensor [n, max_len]."""
        if isinstance(x_dg, t.Tensor):
            return x_dg.long()
        lens = [len(sentence) for sentence in x_dg]
        max_seq_len = int(np.amax(lens))
        pad = self.gen_batch_loader.word_to_idx[self.gen_batch_loader.pad_token]
        out = [list(line) + [pad] * (max_seq_len - n) for line, n in zip(x_dg, lens)]
        return t.from_numpy(np.array(out, dtype=np.int64).reshape((len(out), max_seq_len)))
