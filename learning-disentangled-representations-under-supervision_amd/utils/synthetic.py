"""Synthetic data with the layout of the reference's loaders (utils/batch_loader.py:271-316, utils/data_handler.py:69-102)
for tests and benchmarks: fixed-length sentences, encoder input reversed, dec = [go]+s, target = s+[end], the three
special tokens are the last vocabulary indices (batch_loader.py:81-85). The real text pipeline (pickled corpora,
vocabulary building) is CPU data preparation outside the hot path."""
import numpy as np
import torch


def word_embeddings(config, seed=1234):
    g = torch.Generator().manual_seed(seed)
    return ((torch.rand(config.word_vocab_size, config.word_embed_size, generator=g) * 2 - 1) * 0.25).numpy()


class _GenBatchLoader:
    def __init__(self, config, L, train_lines, seed, val_lines=0):
        self.config, self.L, self.train_lines, self.seed = config, L, train_lines, seed
        self.val_lines = val_lines
        # special tokens are the last three vocabulary indices (batch_loader.py:81-85, :170)
        V = config.word_vocab_size
        self.pad_token, self.go_token, self.end_token = '_', '>', '|'
        self.word_to_idx = {self.pad_token: V - 3, self.go_token: V - 2, self.end_token: V - 1}

    def decode_word(self, idx):
        inv = {v: k for k, v in self.word_to_idx.items()}
        return inv.get(int(idx), 'w%d' % int(idx))

    def next_batch(self, batch_size, target_str, start_index):
        """Five arrays like batch_loader.py:271-316: [enc_word, enc_char, dec_word, dec_char, target] (int64)."""
        cfg, L = self.config, self.L
        V = cfg.word_vocab_size
        rng = np.random.RandomState((self.seed + 7919 * start_index + (0 if target_str == 'train' else 1)) % (2 ** 31))
        s = rng.randint(0, V - 3, size=(batch_size, L)).astype(np.int64)
        chars = rng.randint(0, cfg.char_vocab_size, size=(batch_size, L, cfg.max_word_len)).astype(np.int64)
        go, end = V - 2, V - 1
        enc_word = np.ascontiguousarray(s[:, ::-1])
        enc_char = np.ascontiguousarray(chars[:, ::-1])
        dec_word = np.concatenate([np.full((batch_size, 1), go, np.int64), s], 1)
        dec_char = np.zeros((batch_size, L + 1, cfg.max_word_len), np.int64)
        target = np.concatenate([s, np.full((batch_size, 1), end, np.int64)], 1)
        return [enc_word, enc_char, dec_word, dec_char, target]


class SyntheticDataHandler:
    """Duck-types the parts of utils/data_handler.py DataHandler that the step closures touch."""

    def __init__(self, config, batch_size, seq_len, num_batches=8, seed=1234, val_lines=0, num_dev_batches=None):
        self.config, self.batch_size, self.L, self.seed = config, batch_size, seq_len, seed
        self.num_train_sentiment_batches = num_batches
        self.num_dev_sentiment_batches = num_batches if num_dev_batches is None else num_dev_batches
        self.gen_batch_loader = _GenBatchLoader(config, seq_len, batch_size * num_batches, seed, val_lines)

    def _labelled(self, batch_index, salt):
        rng = np.random.RandomState((self.seed + 104729 * batch_index + salt) % (2 ** 31))
        X = rng.randint(0, self.config.word_vocab_size - 3, size=(self.batch_size, self.L)).tolist()
        Y = rng.randint(0, 2, size=(self.batch_size,)).tolist()
        return X, Y

    def get_sentiment_train_batch(self, batch_index):
        return self._labelled(batch_index, 11)

    def get_sentiment_dev_batch(self, batch_index):
        return self._labelled(batch_index, 13)

    def create_sentiment_batch(self, sentences):
        """data_handler.py:87-102: pad to the in-batch maximum with the pad token (third-last vocab index)."""
        if isinstance(sentences, torch.Tensor):
            return sentences.long()
        pad = self.config.word_vocab_size - 3
        m = max(len(s) for s in sentences)
        out = np.full((len(sentences), m), pad, np.int64)
        for i, s in enumerate(sentences):
            out[i, :len(s)] = s
        return torch.from_numpy(out)

    def synthetic_generated_samples(self, n, step=0):
        rng = np.random.RandomState((self.seed + 31 * step + 17) % (2 ** 31))
        X = rng.randint(0, self.config.word_vocab_size - 3, size=(n, self.L)).tolist()
        c = torch.from_numpy(rng.randint(0, 2, size=(n, 1)).astype(np.float32))
        return X, c


# ------------------------------------------------------------------------------------------------------------------
# Seeded tensors of one benchmark / parity batch. The CPU oracle (test infrastructure) has generators of
# its own that draw the same streams; bench.py's measured leg uses THESE, so that nothing measured imports the oracle.
# ------------------------------------------------------------------------------------------------------------------
def synthetic_word_embed(config, seed=1234):
    """Frozen word embedding ~ U(-1,1) * 0.25, [V, E] float32 numpy."""
    g = torch.Generator().manual_seed(seed)
    return ((torch.rand(config.word_vocab_size, config.word_embed_size, generator=g) * 2 - 1) * 0.25).numpy()


def synthetic_batch(config, B, L, seed=1234):
    """Fixed-length sentences laid out as utils/batch_loader.py:271-316 does (enc_* reversed, dec = [go]+s,
    target = s+[end], special tokens = the last vocabulary indices) plus the step's random draws and the two
    discriminator batches, as CPU tensors."""
    g = torch.Generator().manual_seed(seed)
    V = config.word_vocab_size
    s = torch.randint(0, V - 3, (B, L), generator=g)
    chars = torch.randint(0, config.char_vocab_size, (B, L, config.max_word_len), generator=g)
    go, end = V - 2, V - 1
    out = dict(
        enc_word=torch.flip(s, dims=[1]).contiguous(), enc_char=torch.flip(chars, dims=[1]).contiguous(),
        dec_word=torch.cat([torch.full((B, 1), go, dtype=torch.int64), s], 1),
        target=torch.cat([s, torch.full((B, 1), end, dtype=torch.int64)], 1),
        eps=torch.randn(B, config.latent_variable_size, generator=g),
        eps_recon=torch.randn(B, config.latent_variable_size, generator=g),
        c_uniform=torch.rand(B, generator=g),
        x_labelled=torch.randint(0, V - 3, (B, L), generator=g),
        y_labelled=torch.randint(0, 2, (B,), generator=g),
        x_generated=torch.randint(0, V - 3, (B, L), generator=g),
        c_generated=torch.randint(0, 2, (B,), generator=g),
    )
    C = getattr(config, "c_size", 1)
    if C > 1:  # multi-attribute extension: one label / code / uniform column per attribute
        for k, draw in (("c_uniform", lambda: torch.rand(B, generator=g)),
                        ("y_labelled", lambda: torch.randint(0, 2, (B,), generator=g)),
                        ("c_generated", lambda: torch.randint(0, 2, (B,), generator=g))):
            out[k] = torch.stack([out[k]] + [draw() for _ in range(C - 1)], 1)
    return out

