"""Sleep-phase beam sampler on the MI355X: model/cgn.py:747-921 (sample_from_generator) with utils/beam_search.py
(Beam) and model/generator.py:24-43 (only_decoder_beam) of the reference, batched over all sentences and beams.

The reference drives one Python `Beam` per sentence: per token it stacks the beams' words, runs one LSTM step with
carried state, takes log_softmax(fc(h)), calls Beam.advance (a top-k per sentence on the host side of the loop) and
re-orders every sentence's LSTM state with index_select, then compacts finished sentences away. Here one decode step is
a fixed sequence of kernels over ALL S*K rows (row r = s*K + k):
    word rows gather -> layer-0 gate GEMM (+ the [z,c] contribution, computed once per call) -> LSTM step per layer
    (state carried in [R, 2, H] history buffers) -> vocabulary GEMM -> row log-sum-exp -> cgn_beam_topk_f32 (top-K per
    sentence, scores, back-pointers, done flags) -> state re-ordering as row gathers driven by the kernel's src_rows.
Finished sentences are frozen by the kernel instead of being compacted away, so every sentence keeps its own seed:
the reference's seed tiling (generator.py:30-32) pairs the remaining sentences with the wrong seeds once one has
finished; until then the two are identical. The host reads the done flags once per token (the
reference syncs per token as well) and walks the back-pointers at the end (Beam.get_hyp, beam_search.py:121-132).
"""
import torch

from . import engine, ops
from ._lib import Axis

LIN = Axis(1, 0, 0)


def _new(shape, dev, dtype=torch.float32):
    return ops.alloc(shape, dev, dtype)


class BeamSampler:
    def __init__(self, eng):
        self.eng = eng

    def sample(self, seed, seq_len, beam_size, go, eos, pad, n_best=1):
        """seed [S, Z+C] (device fp32). Returns (hyps, scores): hyps[s] = list over n_best of token lists,
        scores[s] = the final beam scores of sentence s, descending (device tensor [K])."""
        eng = self.eng
        cfg, P, dev = eng.cfg, eng.P, eng.dev
        E, H, V = cfg.word_embed_size, cfg.decoder_rnn_size, cfg.word_vocab_size
        nl = cfg.decoder_num_layers
        S, K = seed.shape[0], int(beam_size)
        R = S * K
        if seed.shape[1] != cfg.latent_variable_size + cfg.c_size:
            raise ValueError("seed must be [S, Z + C]")
        table = P["embedding.word_embed.weight"]
        Wih0 = P["generator.rnn.weight_ih_l0"]
        # the [z,c] part of the layer-0 gates: once per call, then one row per beam
        zg = _new((S, 4 * H), dev)
        ops.gemm(seed.contiguous(), Wih0[:, E:].t(), zg)
        sent_of_row = torch.arange(R, device=dev, dtype=torch.int64) // K
        zgK = _new((R, 4 * H), dev)
        ops.gather_rows(zg, sent_of_row, zgK)

        cur = torch.full((R,), int(pad), dtype=torch.int64, device=dev)  # beam_search.py:37-38
        cur[::K] = int(go)
        scores = torch.zeros(R, dtype=torch.float32, device=dev)  # :31
        done = torch.zeros(S, dtype=torch.int32, device=dev)
        length = torch.zeros(S, dtype=torch.int32, device=dev)
        origin = torch.zeros(R, dtype=torch.int32, device=dev)
        src_rows = torch.zeros(R, dtype=torch.int64, device=dev)
        hist_tok = torch.zeros(seq_len, R, dtype=torch.int64, device=dev)
        hist_org = torch.zeros(seq_len, R, dtype=torch.int32, device=dev)
        zero_tgt = torch.zeros(R, dtype=torch.int64, device=dev)
        state = [None] * nl  # per layer (h [R,H], c [R,H]) contiguous; None = zero state

        for i in range(seq_len):  # cgn.py:786
            emb = _new((R, E), dev)
            ops.gather_rows(table, cur, emb)  # cgn.py:793
            x_op = None
            seqs = []
            for l in range(nl):  # generator.py:41, one timestep with carried state
                sq = engine.LstmSeq(R, 1, H, dev, False)
                sfx = "_l%d" % l
                W_ih = P["generator.rnn.weight_ih" + sfx]
                if l == 0:
                    ops.gemm(emb, W_ih[:, :E].t(), sq.xg.view(R, 4 * H), bias0=P["generator.rnn.bias_ih" + sfx],
                             bias1=P["generator.rnn.bias_hh" + sfx])
                    sq.xg_extra = zgK
                else:
                    hp, hax = x_op
                    ops.gemm(hp, W_ih.t(), sq.xg.view(R, 4 * H), M=R, K=H, a_m=hax, a_k=LIN,
                             bias0=P["generator.rnn.bias_ih" + sfx], bias1=P["generator.rnn.bias_hh" + sfx])
                if state[l] is None:
                    sq.set_initial()
                else:
                    sq.set_initial(state[l][0], state[l][1])
                engine.lstm_run_fwd([sq], [P["generator.rnn.weight_hh" + sfx]])
                x_op = sq.hout_operand()
                seqs.append(sq)
            hp, hax = x_op
            logits = _new((R, V), dev)
            ops.gemm(hp, P["generator.fc.weight"].t(), logits, M=R, K=H, a_m=hax, a_k=LIN,
                     bias0=P["generator.fc.bias"])  # cgn.py:803
            lse, _, _ = ops.softmax_nll_fwd(logits, zero_tgt, 1.0)  # log_softmax = logits - lse
            ops.beam_topk(logits, lse, S, K, i == 0, eos, pad, scores, done, length, cur, origin, src_rows,
                          hist_tok[i], hist_org[i])
            # cgn.py:823-838: every beam continues from its parent's state (rows of the [R,2,H] buffers, slot 1)
            for l, sq in enumerate(seqs):
                h_new, c_new = _new((R, H), dev), _new((R, H), dev)
                ops.gather_rows(sq.hbuf.view(2 * R, H), src_rows, h_new)
                ops.gather_rows(sq.cbuf.view(2 * R, H), src_rows, c_new)
                state[l] = (h_new, c_new)
            if bool(done.all().item()):  # cgn.py:840-841 (host check per token, as in the reference)
                break

        # (4) package everything up, cgn.py:872-882: walk the back-pointers (Beam.get_hyp, beam_search.py:121-132)
        tok_h, org_h, len_h = hist_tok.cpu(), hist_org.cpu(), length.cpu().tolist()
        sc = scores.view(S, K)
        order = torch.sort(sc, 1, True)[1].cpu()  # sort_best (:101-103); already descending up to ties
        hyps = []
        for s in range(S):
            out = []
            for k in order[s, :n_best].tolist():
                hyp = []
                for j in range(len_h[s] - 1, -1, -1):
                    hyp.append(int(tok_h[j, s * K + k]))
                    k = int(org_h[j, s * K + k])
                out.append(hyp[::-1])
            hyps.append(out)
        return hyps, [torch.sort(sc[s], 0, True)[0] for s in range(S)]
