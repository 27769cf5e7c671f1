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


class _DecodeState:
    """Static buffers of one (S, K) decode loop, so that a token step can be captured once and replayed."""

    def __init__(self, eng, S, K):
        cfg, dev = eng.cfg, eng.dev
        H, nl = cfg.decoder_rnn_size, cfg.decoder_num_layers
        R = S * K
        self.S, self.K, self.R = S, K, R
        z = lambda *shape, dt=torch.float32: torch.zeros(*shape, dtype=dt, device=dev)  # noqa: E731
        self.zgK = z(R, 4 * H)
        self.cur = z(R, dt=torch.int64)
        self.scores = z(R)
        self.done, self.length = z(S, dt=torch.int32), z(S, dt=torch.int32)
        self.origin, self.src_rows = z(R, dt=torch.int32), z(R, dt=torch.int64)
        self.tok_out, self.org_out = z(R, dt=torch.int64), z(R, dt=torch.int32)
        self.zero_tgt = z(R, dt=torch.int64)
        self.h = [z(R, H) for _ in range(nl)]
        self.c = [z(R, H) for _ in range(nl)]
        self.sent_of_row = torch.arange(R, device=dev, dtype=torch.int64) // K
        self.graphs = None  # (first-step graph, next-step graph), built on first use
        self.keep = None

    def reset(self, go, pad):
        self.cur.fill_(int(pad))  # beam_search.py:37-38
        self.cur[::self.K] = int(go)
        for t_ in [self.scores, self.done, self.length] + self.h + self.c:
            t_.zero_()


class BeamSampler:
    """use_graph: capture the token step (first / subsequent) into two hipGraphs per (S, K) and replay them; the
    eager form issues the same kernels one by one."""

    def __init__(self, eng, use_graph=True):
        self.eng = eng
        self.use_graph = use_graph
        # Decode states (static buffers + captured token-step graphs, which hold raw pointers into THIS engine's
        # parameter arenas) are kept on the engine object: they die with the arenas they point into, and a new engine
        # (model.cuda() / set_data_parallel() build one) can never pick up another engine's graphs.
        if not hasattr(eng, "_decode_states"):
            eng._decode_states = {}

    def _token_step(self, st, first, eos, pad):
        """One decode step over all S*K rows, reading and updating the static state in `st`."""
        eng = self.eng
        cfg, P, dev = eng.cfg, eng.P, eng.dev
        E, H, V = cfg.word_embed_size, cfg.decoder_rnn_size, cfg.word_vocab_size
        nl, R, S, K = cfg.decoder_num_layers, st.R, st.S, st.K
        emb = _new((R, E), dev)
        ops.gather_rows(P["embedding.word_embed.weight"], st.cur, emb)  # cgn.py:793
        x_op = None
        seqs = []
        for l in range(nl):  # generator.py:41, one timestep with carried state
            sq = engine.LstmSeq(R, 1, H, dev, False)
            sfx = "_l%d" % l
            W_ih = P["generator.rnn.weight_ih" + sfx]
            if l == 0:
                ops.gemm(emb, W_ih[:, :E].t(), sq.xg.view(R, 4 * H), bias0=P["generator.rnn.bias_ih" + sfx],
                         bias1=P["generator.rnn.bias_hh" + sfx])
                sq.xg_extra = st.zgK
            else:
                hp, hax = x_op
                ops.gemm(hp, W_ih.t(), sq.xg.view(R, 4 * H), M=R, K=H, a_m=hax, a_k=LIN,
                         bias0=P["generator.rnn.bias_ih" + sfx], bias1=P["generator.rnn.bias_hh" + sfx])
            sq.set_initial(st.h[l], st.c[l])
            engine.lstm_run_fwd([sq], [P["generator.rnn.weight_hh" + sfx]])
            x_op = sq.hout_operand()
            seqs.append(sq)
        hp, hax = x_op
        logits = _new((R, V), dev)
        ops.gemm(hp, P["generator.fc.weight"].t(), logits, M=R, K=H, a_m=hax, a_k=LIN,
                 bias0=P["generator.fc.bias"])  # cgn.py:803
        lse, _, _ = ops.softmax_nll_fwd(logits, st.zero_tgt, 1.0)  # log_softmax = logits - lse
        st.step_out = (logits, lse)  # learning=True keeps every step's rows (static addresses inside a captured step)
        ops.beam_topk(logits, lse, S, K, first, eos, pad, st.scores, st.done, st.length, st.cur, st.origin,
                      st.src_rows, st.tok_out, st.org_out)
        # cgn.py:823-838: every beam continues from its parent's state (rows of the [R,2,H] buffers, slot 1)
        for l, sq in enumerate(seqs):
            ops.gather_rows(sq.hbuf.view(2 * R, H), st.src_rows, st.h[l])
            ops.gather_rows(sq.cbuf.view(2 * R, H), st.src_rows, st.c[l])

    def _build_graphs(self, st, go, eos, pad):
        dev = self.eng.dev
        stream = torch.cuda.Stream(dev)
        stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(stream):  # warm-up: kernel loading and per-stream workspaces before capture
            st.reset(go, pad)
            self._token_step(st, True, eos, pad)
            self._token_step(st, False, eos, pad)
        torch.cuda.current_stream(dev).wait_stream(stream)
        torch.cuda.synchronize(dev)
        from .. import parallel
        parallel.quiesce_before_capture(dev)  # (RCCL watchdog vs. capture: see there)
        graphs, outs = [], []
        prev_keep, ops.KEEP = ops.KEEP, []
        try:
            with ops.capture_without_gc():
                for first in (True, False):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=stream):
                        self._token_step(st, first, eos, pad)
                    graphs.append(g)
                    outs.append(st.step_out)
            st.keep = ops.KEEP
        finally:
            ops.KEEP = prev_keep
        st.graphs, st.graph_outs = tuple(graphs), tuple(outs)
        st.tokens_key = (go, eos, pad)

    def sample(self, seed, seq_len, beam_size, go, eos, pad, n_best=1, learning=False):
        """seed [S, Z+C] (device fp32). Returns (hyps, scores): hyps[s] = list over n_best of token lists,
        scores[s] = the final beam scores of sentence s, descending (device tensor [K]).
        learning=True (cgn.py:880,901-902): returns instead, per sentence, the [len_s, V] device tensor of log_softmax
        rows that Beam.get_hyp_probs (beam_search.py:140-152) collects: walking the back-pointers from beam 0, the
        distribution of the PARENT beam at every step. Every step's [S*K, V] logits are kept for the walk-back
        (seq_len * S * K * V * 4 bytes: 160 MB for 10 sentences at BASELINE configs[1])."""
        eng = self.eng
        cfg, P, dev = eng.cfg, eng.P, eng.dev
        E, H = cfg.word_embed_size, cfg.decoder_rnn_size
        S, K = seed.shape[0], int(beam_size)
        R = S * K
        if seed.shape[1] != cfg.latent_variable_size + cfg.c_size:
            raise ValueError("seed must be [S, Z + C]")
        key = (S, K)
        st = eng._decode_states.get(key)
        if st is None:
            if len(eng._decode_states) >= 4:  # bounded: each state holds an [S*K, V] logits buffer and two graphs
                eng._decode_states.pop(next(iter(eng._decode_states)))
            st = eng._decode_states[key] = _DecodeState(eng, S, K)
        graphed = self.use_graph and ops.KEEP is None  # never nest inside another capture
        if graphed and (st.graphs is None or st.tokens_key != (go, eos, pad)):
            self._build_graphs(st, go, eos, pad)
        # the [z,c] part of the layer-0 gates: once per call, then one row per beam
        zg = _new((S, 4 * H), dev)
        ops.gemm(seed.contiguous(), P["generator.rnn.weight_ih_l0"][:, E:].t(), zg)
        ops.gather_rows(zg, st.sent_of_row, st.zgK)
        st.reset(go, pad)
        hist_tok = torch.zeros(seq_len, R, dtype=torch.int64, device=dev)
        hist_org = torch.zeros(seq_len, R, dtype=torch.int32, device=dev)
        if learning:
            V = cfg.word_vocab_size
            hist_logits, hist_lse = _new((seq_len * R, V), dev), _new((seq_len * R,), dev)
        for i in range(seq_len):  # cgn.py:786
            if graphed:
                st.graphs[0 if i == 0 else 1].replay()
            else:
                self._token_step(st, i == 0, eos, pad)
            hist_tok[i].copy_(st.tok_out, non_blocking=True)
            hist_org[i].copy_(st.org_out, non_blocking=True)
            if learning:  # beam_search.py:73 `self.output_probs.append(workd_lk)`
                lg, ls = st.graph_outs[0 if i == 0 else 1] if graphed else st.step_out
                hist_logits[i * R:(i + 1) * R].copy_(lg, non_blocking=True)
                hist_lse[i * R:(i + 1) * R].copy_(ls, non_blocking=True)
            if bool(st.done.all().item()):  # cgn.py:840-841 (host check per token, as in the reference)
                break
        ops.raise_on_lstm_timeout()  # the loop has just synchronised: one more word read costs nothing

        # (4) package everything up, cgn.py:872-882: walk the back-pointers (Beam.get_hyp, beam_search.py:121-132)
        tok_h, org_h, len_h = hist_tok.cpu(), hist_org.cpu(), st.length.cpu().tolist()
        if learning:
            picks = []
            for s in range(S):
                k, rows = 0, []
                for j in range(len_h[s] - 1, -1, -1):  # beam_search.py:145-150: the parent first, then its row
                    k = int(org_h[j, s * K + k])
                    rows.append(j * R + s * K + k)
                picks.append(rows[::-1])
            flat = torch.tensor([r for rows in picks for r in rows], dtype=torch.int64, device=dev)
            lp = ops.log_softmax_rows(hist_logits, hist_lse, flat)
            return list(torch.split(lp, [len(rows) for rows in picks]))
        sc = st.scores.clone().view(S, K)
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
