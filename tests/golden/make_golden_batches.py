"""Generates tests/golden/reference_batches.npz by RUNNING the reference's own batching code (build container only):
utils/batch_loader.py BatchLoader.build_word_vocab / encode_characters / next_batch / go_input and
utils/data_handler.py DataHandler.create_sentiment_batch. Both files are Python-2 syntax (print statements, cPickle):
they are converted IN MEMORY with lib2to3 and exec'd with a stub `cPickle`; nothing is written to disk and the objects
are built with object.__new__ (their constructors read pickled corpora from hard-coded paths). The ragged
np.array(...) of create_tensors (batch_loader.py:264-269) is rejected by numpy 2.x, so word_tensor / character_tensor
are built here with the same expressions as plain lists; next_batch only indexes them.

The .npz holds only data: the toy corpus (as word-id arrays), the vocabulary order and the batches the reference built.
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_batches.npz")

CORPUS = ["the movie was great", "i did not like it at all", "a fine film", "what a waste of time , really",
          "great", "the plot was thin but the acting was fine", "not great not bad", "i like it"]


def load(relpath, name, extra=None):
    from lib2to3 import refactor
    src = open(os.path.join(REF, relpath)).read()
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package("lib2to3.fixes"))
    py3 = str(tool.refactor_string(src + "\n", relpath))
    mod = types.ModuleType("utils." + name)
    mod.__package__ = "utils"  # 2to3 turns the sibling imports into relative ones (from .functional import *)
    mod.__dict__.update(extra or {})
    exec(compile(py3, relpath + "(2to3)", "exec"), mod.__dict__)
    return mod


def main():
    sys.path.insert(0, REF)
    sys.modules.setdefault("cPickle", types.ModuleType("cPickle"))
    sys.modules.setdefault("_pickle", sys.modules["cPickle"])
    bl_mod = load("utils/batch_loader.py", "ref_batch_loader")
    BL = bl_mod.BatchLoader
    sys.modules["utils.batch_loader"] = sys.modules["batch_loader"] = bl_mod  # data_handler.py imports it (Python-2 file)
    bl = object.__new__(BL)
    bl.blind_symbol, bl.pad_token, bl.go_token, bl.end_token, bl.a_token = '', '_', '>', '|', '?'  # batch_loader.py:81-85
    words = [w for line in CORPUS for w in line.split()]
    bl.words_vocab_size, bl.idx_to_word, bl.word_to_idx = bl.build_word_vocab(words)  # :163-177
    # build_character_vocab (:142-161) orders characters by set iteration (hash-seed dependent): fix a sorted order
    chars = sorted(set("".join(CORPUS))) + [bl.blind_symbol, bl.pad_token, bl.go_token, bl.end_token]
    bl.chars_vocab_size, bl.idx_to_char, bl.char_to_idx = len(chars), chars, {c: i for i, c in enumerate(chars)}
    bl.max_word_len = int(np.amax([len(w) for w in bl.idx_to_word]))  # :199
    data_words = [[line.split() for line in CORPUS]]
    bl.word_tensor = [[list(map(bl.word_to_idx.get, line)) for line in target] for target in data_words]  # :265-266
    bl.character_tensor = [[list(map(bl.encode_characters, line)) for line in target] for target in data_words]  # :268-269
    bl.train_lines = 6
    out = {"vocab": np.array(bl.idx_to_word), "chars": np.array(chars), "max_word_len": np.array(bl.max_word_len),
           "corpus": np.array(CORPUS)}
    for tag, (bs, split, start) in {"train0": (4, "train", 0), "train1": (3, "train", 3), "valid0": (2, "valid", 0)}.items():
        arrs = bl.next_batch(bs, split, start)  # :271-316
        for nm, a in zip(("enc_word", "enc_char", "dec_word", "dec_char", "target"), arrs):
            out["%s/%s" % (tag, nm)] = np.asarray(a)
    gw, gc = bl.go_input(3)  # :341-345
    out["go_word"], out["go_char"] = gw, gc

    import torch
    dh_mod = load("utils/data_handler.py", "ref_data_handler", {"BatchLoader": BL})
    dh = object.__new__(dh_mod.DataHandler)
    dh.gen_batch_loader = bl
    ragged = [[3, 1, 4], [1], [5, 9, 2, 6, 5], [3, 5]]
    out["sentiment_ragged"] = np.array([len(r) for r in ragged] + [x for r in ragged for x in r])
    out["sentiment_batch"] = dh.create_sentiment_batch([list(r) for r in ragged]).numpy()  # data_handler.py:87-102
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
