"""Vocabulary + batch assembly with the layout of the reference's utils/batch_loader.py, for corpora held in memory.

Same attribute and method names as the reference's BatchLoader for everything the training loop, the sampler and the
step closures touch (word_to_idx / idx_to_word / char_to_idx, the four special tokens, max_word_len, train_lines,
next_batch, go_input, encode_characters, decode_word); the batches are pinned against the reference's own next_batch
(tests/golden/reference_batches.npz). Not rebuilt: the pickled-corpus constructor paths and the hard-coded
train_lines = 900000 (batch_loader.py:87-104, :244) — CPU file handling outside the hot path; build from a list of
sentences instead. One deliberate difference: the character vocabulary is sorted (the reference takes set-iteration
order, batch_loader.py:146-154, which changes from run to run under Python 3 hash randomisation).
"""
import numpy as np


class BatchLoader:
    def __init__(self, sentences, train_lines=None, idx_to_word=None, idx_to_char=None):
        """sentences: list of whitespace-tokenised strings (one corpus, the reference's data[0]).
        train_lines: sentences [0, train_lines) are 'train', the rest 'valid' (batch_loader.py:244-245, :273-274).
        idx_to_word / idx_to_char: optional fixed vocabularies (what load_preprocessed reads from the idx files)."""
        self.blind_symbol = ''   # batch_loader.py:81-85
        self.pad_token = '_'
        self.go_token = '>'
        self.end_token = '|'
        self.a_token = '?'
        data_words = [line.split() for line in sentences]
        if idx_to_word is None:
            merged = [w for line in data_words for w in line]
            self.words_vocab_size, self.idx_to_word, self.word_to_idx = self.build_word_vocab(merged)
        else:
            self.idx_to_word = list(idx_to_word)
            self.words_vocab_size = len(self.idx_to_word)
            self.word_to_idx = {x: i for i, x in enumerate(self.idx_to_word)}
        if idx_to_char is None:
            self.chars_vocab_size, self.idx_to_char, self.char_to_idx = self.build_character_vocab(sentences)
        else:
            self.idx_to_char = list(idx_to_char)
            self.chars_vocab_size = len(self.idx_to_char)
            self.char_to_idx = {x: i for i, x in enumerate(self.idx_to_char)}
        self.max_word_len = int(np.amax([len(word) for word in self.idx_to_word]))  # :199
        self.max_seq_len = int(np.amax([len(line) for line in data_words])) if data_words else 0  # :200
        self.num_lines = [len(data_words)]
        self.train_lines = len(data_words) if train_lines is None else int(train_lines)
        self.val_lines = self.num_lines[0] - self.train_lines
        self.create_tensors([data_words])

    # batch_loader.py:142-161 (sorted, see module docstring)
    def build_character_vocab(self, sentences):
        chars = set()
        for sen in sentences:
            chars = set(sen).union(chars)
        chars = sorted(chars) + [self.blind_symbol, self.pad_token, self.go_token, self.end_token]
        return len(chars), chars, {x: i for i, x in enumerate(chars)}

    # batch_loader.py:163-177
    def build_word_vocab(self, words):
        """Distinct words in sorted order, then the three special tokens as the LAST indices (pad, go, end)."""
        vocab = sorted(set(words)) + [self.pad_token, self.go_token, self.end_token]
        return len(vocab), vocab, dict(zip(vocab, range(len(vocab))))

    # batch_loader.py:264-269 (kept as lists: the reference's ragged np.array is rejected by numpy 2.x)
    def create_tensors(self, data_words):
        self.word_tensor = [[list(map(self.word_to_idx.get, line)) for line in target] for target in data_words]
        self.character_tensor = [[list(map(self.encode_characters, line)) for line in target] for target in data_words]

    # batch_loader.py:271-316
    def next_batch(self, batch_size, target_str, start_index):
        """[encoder_word_input (reversed, LEFT-padded), encoder_character_input, decoder_word_input ([go] + line),
        decoder_character_input, decoder_output (line + [end])], each padded to the longest line of the batch."""
        if target_str == 'valid':
            start_index = start_index + self.train_lines
        target = 0
        indexes = range(start_index, start_index + batch_size)
        pad, go, end = (self.word_to_idx[t] for t in (self.pad_token, self.go_token, self.end_token))
        pad_c = self.encode_characters(self.pad_token)
        words = [list(self.word_tensor[target][i]) for i in indexes]
        chars = [list(self.character_tensor[target][i]) for i in indexes]
        lens = [len(line) for line in words]
        mx = int(np.amax(lens))
        enc_w, enc_c, dec_w, dec_c, dec_o = [], [], [], [], []
        for line, cline, n in zip(words, chars, lens):
            add = mx - n
            dec_w.append([go] + line + [pad] * add)
            dec_c.append([self.encode_characters(self.go_token)] + cline + [pad_c] * add)
            dec_o.append(line + [end] + [pad] * add)
            enc_w.append([pad] * add + line[::-1])
            enc_c.append([pad_c] * add + cline[::-1])
        return [np.array(enc_w), np.array(enc_c), np.array(dec_w), np.array(dec_c), np.array(dec_o)]

    # batch_loader.py:341-345
    def go_input(self, batch_size):
        """One <go> step per sentence: word ids [batch, 1] and their character rows [batch, 1, max_word_len]."""
        word = np.full((batch_size, 1), self.word_to_idx[self.go_token], dtype=np.int64)
        chars = np.tile(np.asarray(self.encode_characters(self.go_token), dtype=np.int64), (batch_size, 1, 1))
        return word, chars

    # batch_loader.py:352-370
    def decode_word(self, word_idx):
        return self.idx_to_word[int(word_idx)]

    def encode_characters(self, characters):
        """Character ids of one word, right-padded with the blind symbol to max_word_len."""
        ids = [self.char_to_idx[ch] for ch in characters]
        return ids + [self.char_to_idx[self.blind_symbol]] * (self.max_word_len - len(ids))

    def decode_characters(self, characters_idx):
        return ''.join(self.idx_to_char[i] for i in characters_idx)
