"""The fused CGN wake-sleep training step on MI355X: discriminator step, c inference, encoder/generator step.

This is the hot path named by BASELINE.json. It reproduces, without autograd, the exact gradient routing of the
reference's step closures (model/cgn.py:269-341 and :1049-1154):
  * z is a detached leaf: the generator-side losses reach the encoder only through g = d(79*CE + recon)/dz;
  * the CE and reconstruction backward passes of the reference are summed into ONE pass over the logits
    (dlogits = CE part + soft-embedding part, fused into the dOut·Eᵀ GEMM epilogue, written in place over the logits);
  * the generator's Adam step happens before the encoder's backward; the discriminator is frozen in this step and
    only passes gradients through to the soft embeddings; the word-embedding table never gets a gradient;
  * encoder gradients = d(kld_coef*KLD)/dθ + (dz_out/dθ)ᵀ g.
Parameter gradients land in the flat arenas (hip/arena.py); one Adam kernel per optimizer group.
"""
import contextlib
import os

import torch

from . import engine, ops
from ._lib import ACT_RELU, Axis
from .arena import ParamArena

LIN = Axis(1, 0, 0)

# experiment knobs for the step's tail (where three streams share the GPU): CGN_TAIL_SPLITK / CGN_TAIL_TILE force the
# plan of every product issued there; CGN_HW_DW_EARLY=1 forks the highway's dW products where their inputs are ready
TAIL_HINT = {k: int(os.environ[e]) for k, e in (("splitk", "CGN_TAIL_SPLITK"), ("tile", "CGN_TAIL_TILE"))
             if os.environ.get(e)}
HW_DW_EARLY = os.environ.get("CGN_HW_DW_EARLY", "1") != "0"
# sibling products (the encoder's two W_ih products, the per-direction / per-layer LSTM weight-gradient products) as
# grouped X6 launches (cgn_gemm_group_f32); CGN_GROUP_SIBLINGS=0: one launch per product (A/B switch)
GROUP_SIBLINGS = os.environ.get("CGN_GROUP_SIBLINGS", "1") != "0"
# ... and the weight-gradient products in the step's tail too. Off: measured +35 us per step (2.56 against 2.525 ms, same
# box, profiles/r03_group_ab.txt) - one 400-block launch per stream shares the chip worse with the other two streams'
# work than a few 128-block launches do
GROUP_TAIL = os.environ.get("CGN_GROUP_TAIL", "0") == "1"
SOFTEMB_TILE = int(os.environ.get("CGN_SOFTEMB_TILE", "0"))
SOFTEMB_SPLITK = int(os.environ.get("CGN_SOFTEMB_SPLITK", "0"))


def _new(shape, dev, dtype=torch.float32):
    return ops.alloc(shape, dev, dtype)


def _cap_sel(capture, key, t):
    """Parity-test aid: which ReLU gates were open / which window won each max-over-time in THIS pass (the step is
    piecewise smooth; tests/test_step_gpu.py compares gradients on the region the fp32 pass actually took)."""
    if capture is not None:
        capture.setdefault("sel", {})[key] = t.clone()


class CGNStep:
    def __init__(self, model, allreduce=None, world_size=1, force_dp=False):
        """model: Controlled_Generation_Sentence on a HIP device. allreduce: parallel.BucketedAllReduce-like object
        (begin(flat) -> handles, wait(handles)); each optimizer group's gradient arena is summed over the ranks between
        its last gradient kernel and its Adam kernel (data parallelism), and scaled by 1/world_size inside the Adam
        kernel. force_dp keeps that schedule on for world_size 1 (single-GPU rehearsal)."""
        self.model = model
        self.cfg = model.config
        named = dict(model.named_parameters())
        dev = named["e2mu.weight"].device
        if dev.type != "cuda":
            raise RuntimeError("CGNStep needs the model on an MI355X (HIP) device: call model.cuda() first")
        self.dev = dev
        groups = {"disc": [], "gen": [], "enc": []}
        for n, p in named.items():
            if n == "embedding.word_embed.weight":
                continue  # frozen constant (selfModules/embedding.py:23)
            if n.startswith("sentiment_discriminator.") or n.startswith("attribute_discriminators."):
                groups["disc"].append((n, p))
            elif n.startswith("generator."):
                groups["gen"].append((n, p))
            else:
                groups["enc"].append((n, p))  # cgn.py:676-683: encoder + embedding + e2mu + e2logvar
        self.named_groups = groups
        # weights that always multiply the same activations are stacked in the arena (one GEMM instead of one each)
        self.n_hw = model.encoder.hw1.num_layers
        fused_enc = {}
        for l in range(self.n_hw):
            pre = "encoder.hw1."
            fused_enc["hw%d.w" % l] = [pre + "%s_module_%d-0" % (k, l) for k in ("nonlinear", "linear", "gate")]
            fused_enc["hw%d.b" % l] = [pre + "%s_module_%d-1" % (k, l) for k in ("nonlinear", "linear", "gate")]
        if model.config.latent_variable_size % 4 == 0:
            fused_enc["e2.w"] = ["e2mu.weight", "e2logvar.weight"]
            fused_enc["e2.b"] = ["e2mu.bias", "e2logvar.bias"]
        self.arenas = {k: ParamArena(v, dev, fused=fused_enc if k == "enc" else None) for k, v in groups.items()}
        for k, a in self.arenas.items():
            a.group = k
        self.P = {n: p.data for n, p in named.items()}
        self.G = {}
        for a in self.arenas.values():
            self.G.update(a.grads)
        self.allreduce = allreduce
        self.world_size = world_size
        self.dp = allreduce is not None and (world_size > 1 or force_dp)
        self._pending = {}     # optimizer group -> handles of its gradient all-reduce in flight
        # The encoder group's gradients become final in two waves: the biLSTM's and the mu|logvar head's right behind the
        # LSTM backward (side stream 0), the highway's, the TDNN's and the char table's only at the very end of the step.
        # Its arena is laid out in that order (registration order: embedding, highway, rnn, e2*), so the all-reduce is
        # split at the first LSTM weight: "enc.a" (the late-registered, EARLY-final 22 MB at cfg2) travels beside the dX
        # chain, only "enc.b" (16 MB) is left for the end of the step.
        a = self.arenas["enc"]
        cut = a.offsets["encoder.rnn.weight_ih_l0"][0] if "encoder.rnn.weight_ih_l0" in a.offsets else a.size
        early = [n for n, (o, _, _, _) in a.offsets.items() if o >= cut]
        ok = all(n.startswith("encoder.rnn.") or n.startswith("e2") for n in early) and 0 < cut < a.size
        self._enc_cut = cut if ok else None
        self._split_enc = False
        self.recorder = None   # hip/graph.py: while set, every communication point ends one graph segment
        cfg = self.cfg
        # Stream overlap (off by default: in eager mode the host, not the GPU, is the bound and every fork/join costs two
        # more host calls; hip/graph.py turns it on while capturing the step into a hipGraph). side(k) regions run on
        # side stream k, ordered after everything issued so far on the current stream; join(k) orders the current
        # stream after side stream k.
        self.overlap = False
        self._side_streams = None
        self._dirty = set()
        self._deferring = False
        self._deferred = []
        # Device-resident step scalars {1/tau, 79/(tau*B*T), kld_coef/B, kld_coef}: when `dynamic` the kernels read these
        # instead of immediates baked into their arguments, and Adam reads a device step count, so that a captured
        # graph can be replayed while the schedules (utils/functional.py:28-42) advance.
        self.dynamic = False
        self.dyn = torch.zeros(4, dtype=torch.float32, device=dev)
        self._zero1 = torch.zeros(1, dtype=torch.float32, device=dev)  # read-only
        # one text-CNN per attribute of c (the reference: one, `sentiment_discriminator`; BASELINE configs[4]: two)
        self.disc_prefixes = ["sentiment_discriminator."] + ["attribute_discriminators.%d." % i
                                                             for i in range(len(model.attribute_discriminators))]
        if len(self.disc_prefixes) != cfg.c_size:
            raise ValueError("config.c_size = %d but the model has %d discriminators" % (cfg.c_size,
                                                                                        len(self.disc_prefixes)))
        self.Ks = list(cfg.sentiment_kernel_size)
        self.Cos = [cfg.sentiment_kernel_num] * len(self.Ks)

    def valid(self):
        return all(self.arenas[k].owns(v) for k, v in self.named_groups.items())

    # -------------------------------------------------------------------------------------------------------------
    @contextlib.contextmanager
    def side(self, k):
        if not self.overlap:
            yield
            return
        if self._side_streams is None:
            self._side_streams = [torch.cuda.Stream(self.dev) for _ in range(2)]
        st = self._side_streams[k]
        rec = self.recorder
        if rec is not None and getattr(rec, "parallel", False):
            # recording the step as one hipGraph PER BRANCH (hip/graph.py _BranchRecorder): the region becomes a graph
            # of its own, replayed on side stream k behind everything the main stream was given so far
            rec.begin_side(k, st)
            self._dirty.add(k)
            try:
                with torch.cuda.stream(st):
                    yield
            finally:
                rec.end_side(k)
            return
        ev = torch.cuda.Event()
        ev.record()
        st.wait_event(ev)
        self._dirty.add(k)
        with torch.cuda.stream(st):
            yield

    def join(self, *ks):
        rec = self.recorder
        for k in ks:
            if k in self._dirty:
                if rec is not None and getattr(rec, "parallel", False):
                    rec.join(k)  # the main branch is cut here: what follows waits for side stream k
                else:
                    ev = torch.cuda.Event()
                    ev.record(self._side_streams[k])
                    torch.cuda.current_stream(self.dev).wait_event(ev)
                self._dirty.discard(k)

    # Scheduling of the overlapped (captured) step. Measured on MI355X (profiles/r02_*): a persistent LSTM launch needs
    # its workgroups resident together and holds most of a CU's registers, so a GEMM that shares the chip with it either
    # keeps its workgroups waiting for CUs (the launch then takes 2-3x as long) or cannot get on a CU at all. So:
    #   * exclusive(): every persistent LSTM launch starts with the side streams joined - it runs alone;
    #   * later(k, fn): parameter-gradient work that nothing on the critical path needs is not forked where its inputs
    #     become ready (that would put it beside the next LSTM launch) but queued, and flush() forks all of it behind
    #     the LAST LSTM launch of the step, spread over the side streams, where it overlaps the dX chain's GEMMs.
    # Eager mode (overlap off) and capture (tests read gradients at fixed points) run later() at once.
    # Data-parallel runs keep this schedule (round 3): each optimizer group's all-reduce is started on the stream that
    # produced the group's last gradient and waited for on that same stream, right in front of the group's Adam kernel
    # (_optim_step), so the generator's gradients travel beside the dX chain and no stream is joined for a collective.
    def exclusive(self):
        if self.overlap:
            self.join(0, 1)

    def later(self, k, fn):
        if self._deferring:
            self._deferred.append((k, fn))
        else:
            with self.side(k):
                fn()

    def begin_defer(self, capture):
        self._deferring = bool(self.overlap and capture is None)

    def flush(self, only=None):
        """Fork the queued work now. only=k: just side stream k's share (the rest stays queued, and later() keeps
        queueing for the other streams until a flush without `only`)."""
        prev_hint = ops.PLAN_HINT
        if self.overlap and TAIL_HINT:
            ops.PLAN_HINT = TAIL_HINT
        try:
            self._flush(only)
        finally:
            ops.PLAN_HINT = prev_hint

    def _flush(self, only):
        todo = [(k, fn) for k, fn in self._deferred if only is None or k == only]
        self._deferred = [(k, fn) for k, fn in self._deferred if not (only is None or k == only)]
        if only is None:
            self._deferring = False
        for k in sorted({k for k, _ in todo}, reverse=True):  # one region (one branch graph) per side stream
            with self.side(k):
                for kk, fn in todo:
                    if kk == k:
                        fn()

    @contextlib.contextmanager
    def keepalive(self):
        """With `overlap` on, temporaries allocated on one stream are read by kernels on another; torch's caching
        allocator hands a block back to its own stream's pool the moment the Python reference drops, without knowing
        about the other stream. So while a step with side streams is being ISSUED every temporary is kept (ops.KEEP),
        and released only after the side streams have been joined into the current stream (from then on any reuse is
        ordered behind every reader). A surrounding KEEP (graph capture, hip/graph.py) is left alone."""
        if not self.overlap or ops.KEEP is not None:
            yield
            return
        ops.KEEP = []
        try:
            yield
            self.join(0, 1)
        finally:
            ops.KEEP = None

    def _dyn(self, lo=0):
        return self.dyn[lo:] if self.dynamic else None

    def set_step_scalars(self, kld_coeff, temperature, n_rows, batch, pinned=None):
        """Host -> device copy of the step scalars (async on the current stream when `pinned` is a pinned [4] tensor)."""
        inv_tau = 1.0 / float(temperature)
        vals = (inv_tau, 79.0 * inv_tau / n_rows, float(kld_coeff) / batch, float(kld_coeff))
        if pinned is None:
            pinned = torch.tensor(vals, dtype=torch.float32)
        else:
            for i, v in enumerate(vals):
                pinned[i] = v
        self.dyn.copy_(pinned, non_blocking=True)

    def new_optimizer(self, group):
        """Fresh Adam state for 'disc' | 'enc' | 'gen' (cgn.py:267, :1046-1047 create torch Adam objects likewise)."""
        return self.arenas[group].new_adam()

    def _adam(self, state):
        state.apply(self.cfg.learning_rate, grad_scale=1.0 / self.world_size, dynamic=self.dynamic)

    def _optim_step(self, state):
        if self.dp:
            g = state.arena.group
            if g == "enc" and self._split_enc:  # "enc.a" was started behind the LSTM backward (_encoder_bwd)
                self._comm([("begin", "enc.b"), ("wait", "enc.a"), ("wait", "enc.b")])
                self._split_enc = False
            else:
                self._comm([("begin", g), ("wait", g)])
        self._adam(state)

    def do_comm(self, kind, group):
        """Run one communication action now: ("begin", g) starts the all-reduce of group g's gradient arena on RCCL's
        stream (ordered after the current stream), ("wait", g) orders the current stream after it."""
        if kind == "begin":
            if group == "enc.a":
                flat = self.arenas["enc"].g[self._enc_cut:]
            elif group == "enc.b":
                flat = self.arenas["enc"].g[:self._enc_cut]
            else:
                flat = self.arenas[group].g
            self._pending[group] = self.allreduce.begin(flat)
        else:
            self.allreduce.wait(self._pending.pop(group))

    def _comm(self, actions):
        """A communication point of the data-parallel schedule, on the CURRENT stream (main or a side stream). Eager:
        run the actions. While a step is being recorded (hip/graph.py) the graph being captured on this stream ends
        here and the actions are attached behind it: a replay launches that graph, runs the actions with this stream
        current, then launches the graph that continues the stream. No other stream is joined."""
        if not self.dp:
            return
        if self.recorder is not None:
            self.recorder.comm(list(actions))
            return
        for kind, group in actions:
            self.do_comm(kind, group)

    def _fused(self, name, table):
        """The stacked parameter (table is self.P) or gradient (self.G) view of an arena group, or None."""
        f = self.arenas["enc"].fused.get(name)
        if f is None:
            return None
        return f[0] if table is self.P else f[1]

    def _hw_layers(self, table):
        pre = "encoder.hw1."
        return [dict(nw=table[pre + "nonlinear_module_%d-0" % l], nb=table[pre + "nonlinear_module_%d-1" % l],
                     lw=table[pre + "linear_module_%d-0" % l], lb=table[pre + "linear_module_%d-1" % l],
                     gw=table[pre + "gate_module_%d-0" % l], gb=table[pre + "gate_module_%d-1" % l],
                     wcat=self._fused("hw%d.w" % l, table), bcat=self._fused("hw%d.b" % l, table))
                for l in range(self.n_hw)]

    def _conv_params(self, table, a=0):
        pre = self.disc_prefixes[a]
        w = [table[pre + "convs1.%d.weight" % i] for i in range(len(self.Ks))]
        b = [table[pre + "convs1.%d.bias" % i] for i in range(len(self.Ks))]
        return w, b

    def _fc_params(self, table, a=0):
        pre = self.disc_prefixes[a]
        return table[pre + "fc1.weight"], table[pre + "fc1.bias"]

    def _tdnn_kernels(self, table):
        return [table["embedding.TDNN.TDNN_kernel-%d" % i] for i in range(len(self.cfg.kernels))]

    # -------------------------------------------------------------------------------------------------------------
    # discriminator forward on token ids (word table lookup + text-CNN features)
    # -------------------------------------------------------------------------------------------------------------
    def _embed_ids(self, ids):
        B, L = ids.shape
        emb = _new((B * L, self.cfg.word_embed_size), self.dev)
        ops.gather_rows(self.P["embedding.word_embed.weight"], ids.reshape(-1), emb)
        return emb

    def _disc_features(self, emb, B, L, a=0):
        E = self.cfg.word_embed_size
        cw, cb = self._conv_params(self.P, a)
        return engine.textcnn_features_fwd(emb, B, L, E, L * E, cw, cb)

    def infer_c(self, enc_word, uniform):
        """model/cgn.py:134-172: c ~ Bernoulli(softmax(D(emb(x)))[:,1]) with the uniform draw given. Returns (c, p1):
        [B] for one attribute (the reference), [B, C] with one column per discriminator otherwise."""
        C = self.cfg.c_size
        B, L = enc_word.shape
        emb = self._embed_ids(enc_word)
        c_all, p_all = _new((B, C), self.dev), _new((B, C), self.dev)
        for a in range(C):
            feat, _ = self._disc_features(emb, B, L, a)
            fw, fb = self._fc_params(self.P, a)
            logits, p1, c = _new((B, 2), self.dev), _new((B,), self.dev), _new((B,), self.dev)
            ops.disc_head(feat, fw, fb, logits, p1=p1)
            u = uniform if C == 1 else uniform[:, a].contiguous()
            ops.bernoulli(p1, u, c)
            if C == 1:
                return c, p1
            ops.copy2d(c.view(B, 1), c_all[:, a:a + 1])
            ops.copy2d(p1.view(B, 1), p_all[:, a:a + 1])
        return c_all, p_all

    def disc_train(self, opt, x_labelled, y_labelled, x_generated, c_generated, mask_labelled=None,
                   mask_generated=None, train=True, capture=None, step=True):
        """One discriminator step, model/cgn.py:269-341. Returns the 5-tuple of :341 as 1-element device tensors
        (summed over the attributes' discriminators when c has more than one). y_labelled / c_generated: [B] int64, or
        [B, C] with one label column per attribute. step=False stops after the gradients (the data-parallel schedule
        reduces them and runs Adam later)."""
        cfg, P, G = self.cfg, self.P, self.G
        E, C = cfg.word_embed_size, cfg.c_size
        embs = [(self._embed_ids(x), x.shape) for x in (x_labelled, x_generated)]  # shared by every attribute
        sums = None
        for a in range(C):
            W, b = self._fc_params(P, a)
            dW, db = self._fc_params(G, a)
            dcw, dcb = self._conv_params(G, a)
            col = (lambda t_: t_) if C == 1 else (lambda t_: t_[:, a].contiguous())
            passes = ((embs[0], col(y_labelled), mask_labelled, 1.0, 0.0),
                      (embs[1], col(c_generated), mask_generated, cfg.lambda_u, cfg.lambda_u * cfg.beta))
            stats_all = []
            for i, ((emb, (B, L)), y, mask, w_ce, w_ent) in enumerate(passes):
                feat, arg = self._disc_features(emb, B, L, a)
                _cap_sel(capture, "disc_%s%d/conv" % ("lg"[i], a), arg)
                logits, dlog, stats = _new((B, 2), self.dev), _new((B, 2), self.dev), _new((2,), self.dev)
                ops.disc_head(feat, W, b, logits, mask=mask, y=y, w_ce=w_ce, w_ent=w_ent, dlogits=dlog, stats=stats)
                stats_all.append(stats)
                if not train:
                    continue
                beta = 0.0 if i == 0 else 1.0
                dfeat = _new(tuple(feat.shape), self.dev)
                ops.disc_head_bwd(feat, W, dlog, mask=mask, beta=beta, dW=dW, db=db, dfeat=dfeat)
                ops.conv_bwd_dw(self.Ks, self.Cos, ACT_RELU, 0, dcw, dcb, dfeat, None, arg, emb, L * E, B, L, E,
                                beta=beta)
            ce_l, ce_g, ent = stats_all[0][0:1], stats_all[1][0:1], stats_all[1][1:2]
            gen_loss = ops.lincomb([(1.0, ce_g), (cfg.beta, ent)])  # :329
            total = ops.lincomb([(1.0, ce_l), (cfg.lambda_u, gen_loss)])  # :334
            cur = (total, ce_l, gen_loss, ce_g, ent)
            sums = cur if sums is None else tuple(ops.lincomb([(1.0, x), (1.0, y_)]) for x, y_ in zip(sums, cur))
        if train:
            if capture is not None:
                capture["disc_grads"] = {n: g.clone() for n, g in self.arenas["disc"].grads.items()}
            if step:
                self._optim_step(opt)
        return sums

    # -------------------------------------------------------------------------------------------------------------
    # encoder forward (embedding -> highway -> biLSTM -> mu/logvar); `x_words` None => word part gathered from ids
    # -------------------------------------------------------------------------------------------------------------
    def _encoder_fwd(self, enc_in, B, L, head=True):
        """head=False: stop behind the biLSTM (the caller runs the fused code head, ops.code_head_fwd)."""
        cfg, P = self.cfg, self.P
        He, Z = cfg.encoder_rnn_size, cfg.latent_variable_size
        N = B * L
        hw_out, hw_saved = engine.highway_fwd(enc_in, self._hw_layers(P))
        ops.stamp("M highway fwd done")
        sf, sr = engine.LstmSeq(B, L, He, self.dev, False), engine.LstmSeq(B, L, He, self.dev, True)
        calls = [(hw_out, P["encoder.rnn.weight_ih_l0" + sfx].t(), s.xg.view(N, 4 * He),
                  dict(bias0=P["encoder.rnn.bias_ih_l0" + sfx], bias1=P["encoder.rnn.bias_hh_l0" + sfx]))
                 for s, sfx in ((sf, ""), (sr, "_reverse"))]
        if GROUP_SIBLINGS and ops.get_compute() == "f32" and ops.GEMM_X6:
            ops.gemm_group(calls)  # both directions in one launch: 160 tiles each would leave a third of the CUs idle
        else:
            for a_, b_, o_, kw in calls:
                ops.gemm(a_, b_, o_, **kw)
        sf.set_initial(), sr.set_initial()
        # No exclusive() here: the only work that can still be running is the tail of the discriminator step / c inference
        # on side stream 0, a chain of kernels with 1-113 workgroups that fit beside the recurrence's 256 (measured: the
        # join cost 54 us per step, the chain's last ~80 us ran with the whole GPU waiting; beside the launch it is free).
        ops.stamp("M encoder W_ih products done")
        engine.lstm_run_fwd([sf, sr], [P["encoder.rnn.weight_hh_l0"], P["encoder.rnn.weight_hh_l0_reverse"]])
        ops.stamp("M encoder LSTM fwd done")
        if not head:
            return dict(hw_out=hw_out, hw_saved=hw_saved, sf=sf, sr=sr, ctx=None, mulv=None, B=B, L=L)
        ctx = _new((B, 2 * He), self.dev)
        ops.copy2d(sf.c_final(), ctx[:, :He])
        ops.copy2d(sr.c_final(), ctx[:, He:])
        mulv = _new((B, 2 * Z), self.dev)
        w2 = self._fused("e2.w", P)
        # (compute="f32" in the bf16 mode too: the Mahalanobis z-loss is a difference of two codes (cgn.py:47-57) and
        # amplifies any rounding of mu / logvar; the head is 2*B*2He*2Z FLOP, nothing)
        if w2 is not None:  # [mu | logvar] in one product (cgn.py:91-92)
            ops.gemm(ctx, w2.t(), mulv, bias0=self._fused("e2.b", P), compute="f32")
        else:
            ops.gemm(ctx, P["e2mu.weight"].t(), mulv[:, :Z], bias0=P["e2mu.bias"], compute="f32")
            ops.gemm(ctx, P["e2logvar.weight"].t(), mulv[:, Z:], bias0=P["e2logvar.bias"], compute="f32")
        return dict(hw_out=hw_out, hw_saved=hw_saved, sf=sf, sr=sr, ctx=ctx, mulv=mulv, B=B, L=L)

    def _encoder_bwd(self, es, dmulv, param_grads, dc=None):
        """Backward of _encoder_fwd from d[mu|logvar]. param_grads True: write encoder parameter grads (G);
        returns d(enc_in) [N, Din]. dc = (dcf, dcr): the final cell states' gradients are already there (fused code
        head, ops.code_head_bwd); only the mu|logvar weight gradients are formed from dmulv then."""
        cfg, P, G = self.cfg, self.P, self.G
        He, Z = cfg.encoder_rnn_size, cfg.latent_variable_size
        B, L = es["B"], es["L"]
        N = B * L
        ctx = es["ctx"]
        dctx = _new((B, 2 * He), self.dev) if dc is None else None
        dmu, dlv = dmulv[:, :Z], dmulv[:, Z:]
        w2 = self._fused("e2.w", P)
        if w2 is not None:
            if dc is None:
                ops.gemm(dmulv, w2, dctx, compute="f32")
            if param_grads:
                self.later(0, lambda: engine.linear_bwd(ctx, w2, dmulv, self._fused("e2.w", G), self._fused("e2.b", G)))
        else:
            ops.gemm(dmu, P["e2mu.weight"], dctx, compute="f32")
            ops.gemm(dlv, P["e2logvar.weight"], dctx, beta=1.0, compute="f32")
            if param_grads:
                self.later(0, lambda: (engine.linear_bwd(ctx, P["e2mu.weight"], dmu, G["e2mu.weight"], G["e2mu.bias"]),
                                       engine.linear_bwd(ctx, P["e2logvar.weight"], dlv, G["e2logvar.weight"],
                                                         G["e2logvar.bias"])))
        if dc is None:
            dcf, dcr = _new((B, He), self.dev), _new((B, He), self.dev)
            ops.copy2d(dctx[:, :He], dcf)
            ops.copy2d(dctx[:, He:], dcr)
        else:
            dcf, dcr = dc
        sf, sr = es["sf"], es["sr"]
        self.exclusive()
        dgf, dgr = engine.lstm_run_bwd([sf, sr], [P["encoder.rnn.weight_hh_l0"], P["encoder.rnn.weight_hh_l0_reverse"]],
                                       [None, None], [None, None], [dcf, dcr])
        if param_grads:
            ops.stamp("M encoder LSTM bwd done")
        hw_out = es["hw_out"]
        Din = hw_out.shape[1]
        if param_grads:
            # parameter gradients leave the critical path (dX chain) when streams overlap
            if ops.STAMPS is not None:
                self.later(0, lambda: ops.stamp("S0 deferred work begins (encoder LSTM dW)"))
            items = [(s, dg, (hw_out, Axis(hw_out.stride(0), 0, 0)), G["encoder.rnn.weight_ih_l0" + sfx],
                      G["encoder.rnn.weight_hh_l0" + sfx], G["encoder.rnn.bias_ih_l0" + sfx],
                      G["encoder.rnn.bias_hh_l0" + sfx]) for s, dg, sfx in ((sf, dgf, ""), (sr, dgr, "_reverse"))]
            if GROUP_TAIL and ops.get_compute() == "f32" and ops.GEMM_X6:
                self.later(0, lambda: engine.lstm_param_grads_multi(items))
            else:
                for it in items:
                    self.later(0, lambda it=it: engine.lstm_param_grads(*it))
            # The step's last LSTM launch is behind us: what later() queued forks here. (Inside ONE captured graph a node
            # with three successors - main and both side streams - replayed wrongly on ROCm 7.2: the dX GEMM below ran
            # on unfinished LSTM gradients (tools/graph_diff.py). The in-graph form therefore forks side stream 1 here
            # and side stream 0 one kernel further down; the per-branch graphs of hip/graph.py have no such node.)
            staggered = self.overlap and not getattr(self.recorder, "parallel", False)
            self.flush(only=1 if staggered else None)
        dhw = ops.new2d(N, Din, self.dev)
        if param_grads:
            ops.stamp("M forked deferred work")
            if self.overlap and TAIL_HINT:
                ops.PLAN_HINT = TAIL_HINT  # from here to the end of the step every product shares the GPU
        for i, (dg, sfx) in enumerate(((dgf, ""), (dgr, "_reverse"))):
            ops.gemm(dg.view(N, 4 * He), P["encoder.rnn.weight_ih_l0" + sfx], dhw, beta=(0.0 if i == 0 else 1.0))
            if param_grads and i == 0:
                if ops.STAMPS is not None:
                    self.later(0, lambda: ops.stamp("S0 encoder LSTM dW done"))
                if self._split_enc:  # data parallel: the early-final part of the encoder's gradients starts to travel
                    self.later(0, lambda: self._comm([("begin", "enc.a")]))
                self.flush()
        if param_grads and self.overlap and getattr(self.recorder, "parallel", False) and not HW_DW_EARLY:
            self._deferring = True  # the highway's dW products wait for the end of the dX chain (one more branch graph)
        turn = [0]

        def hw_later(fn):  # the two layers' dW products on different side streams: each one's reduce / column-sum kernels
            self.later(turn[0] % 2, fn)  # run beside the other's GEMM (-0.6 % of the step; an early partial Adam here: +0.5 %)
            turn[0] += 1

        return engine.highway_bwd(es["hw_saved"], self._hw_layers(P), dhw, self._hw_layers(G) if param_grads else None,
                                  need_dx=True, later=hw_later)

    def total_losses(self, ce, kld, c_loss, z_loss, kld_coeff, use_disc=True, calc_z_loss=False):
        """total_encoder_loss, total_generator_loss of cgn.py:1151-1152 from their parts (device scalars)."""
        cfg = self.cfg
        enc_loss = ops.lincomb([(79.0, ce), (float(kld_coeff), kld)])  # cgn.py:1151
        gen_loss = ops.lincomb([(1.0, enc_loss), (cfg.lambda_c if use_disc else 0.0, c_loss),
                                (cfg.lambda_z if calc_z_loss else 0.0, z_loss)])  # cgn.py:1152
        return enc_loss, gen_loss

    def wake_sleep(self, opts, d, kld_coeff, temperature, calc_z_loss=False):
        """One wake-sleep step (train_functions.py:49-108 without the sampler): discriminator step, c inference,
        encoder/generator step. opts: dict(disc, enc, gen); d: dict of device tensors (x_labelled, y_labelled,
        x_generated, c_generated, mask_labelled, mask_generated, enc_word, enc_char, dec_word, target, c_uniform, eps).
        With `overlap` the discriminator step and c inference run beside the encoder forward (they only meet at c)."""
        with self.keepalive():
            return self._wake_sleep(opts, d, kld_coeff, temperature, calc_z_loss)

    def _wake_sleep(self, opts, d, kld_coeff, temperature, calc_z_loss):
        ops.stamp("M step begin")
        with self.side(0):
            ops.stamp("S0 disc step begin")
            d_losses = self.disc_train(opts["disc"], d["x_labelled"], d["y_labelled"], d["x_generated"],
                                       d["c_generated"], d.get("mask_labelled"), d.get("mask_generated"))
            ops.stamp("S0 disc step done")
            c, _ = self.infer_c(d["enc_word"], d["c_uniform"])
            ops.stamp("S0 c inferred")
        eg_losses = self.enc_gen_train((opts["enc"], opts["gen"]), d["enc_word"], d["enc_char"], d["dec_word"],
                                       d["target"], c, d["eps"], kld_coeff, temperature, calc_z_loss=calc_z_loss)
        self.join(0, 1)
        ops.stamp("M step end")
        return d_losses, c, eg_losses

    # -------------------------------------------------------------------------------------------------------------
    def enc_gen_train(self, opts, enc_word, enc_char, dec_word, target, c_target, eps, kld_coeff, temperature,
                      gen_drop_mask=None, calc_z_loss=True, capture=None, train=True, use_disc=True, joint=False):
        """One encoder/generator step, model/cgn.py:1049-1154. All tensor arguments live on the device:
        enc_word [B,L], enc_char [B,L,W], dec_word [B,T], target [B,T] int64; c_target [B] or [B,C] float;
        eps [B,Z] (the randn of cgn.py:96); gen_drop_mask None or [B,T,E] (explicit, scaled word-dropout mask).
        Returns (cross_entropy, kld, z_loss, c_loss, total_encoder_loss, total_generator_loss) device scalars."""
        with self.keepalive():
            return self._enc_gen_train(opts, enc_word, enc_char, dec_word, target, c_target, eps, kld_coeff,
                                       temperature, gen_drop_mask, calc_z_loss, capture, train, use_disc, joint)

    def _enc_gen_train(self, opts, enc_word, enc_char, dec_word, target, c_target, eps, kld_coeff, temperature,
                       gen_drop_mask, calc_z_loss, capture, train, use_disc, joint):
        cfg, P, G, dev = self.cfg, self.P, self.G, self.dev
        B, L = enc_word.shape
        T = dec_word.shape[1]
        E, S = cfg.word_embed_size, cfg.sum_depth
        Din, Hg, Z, C, V = E + S, cfg.decoder_rnn_size, cfg.latent_variable_size, cfg.c_size, cfg.word_vocab_size
        N, NT = B * L, B * T
        Wl = cfg.max_word_len
        inv_tau = 1.0 / float(temperature)
        table = P["embedding.word_embed.weight"]
        if L + 1 != T:
            raise ValueError("generator sequence must be one longer than the encoder's (go/eos token)")
        if L < max(self.Ks):
            raise RuntimeError("Kernel size can't be greater than actual input size")

        # ---- embedding (selfModules/embedding.py:29-55) --------------------------------------------------------
        enc_in = ops.new2d(N, Din, dev)
        ops.gather_rows(table, enc_word.reshape(-1), enc_in[:, :E])
        tk = self._tdnn_kernels(P)
        tfeat, tsaved = engine.tdnn_fwd(enc_char.reshape(N, Wl), P["embedding.char_embed.weight"], tk)
        ops.copy2d(tfeat, enc_in[:, E:])
        _cap_sel(capture, "enc/tdnn", tsaved["arg"])
        ops.stamp("M embedding + TDNN done")
        # ---- encode_for_z (cgn.py:83-103) + KLD (:1076) + [z,c] (:1080) ------------------------------------------
        Wih0 = P["generator.rnn.weight_ih_l0"]
        He = cfg.encoder_rnn_size
        w2f = self._fused("e2.w", P)
        # the code head (context -> mu|logvar -> z, KLD -> [z,c] -> its gate contribution) as ONE launch where the shapes
        # allow (csrc/fused.hip); else the chain of copies, products and the reparametrisation kernel
        fused_head = ops.code_head_eligible(B, He, Z, C, w2f, Wih0[:, E:])
        es = self._encoder_fwd(enc_in, B, L, head=not fused_head)
        if capture is not None:
            for l, (_, pre) in enumerate(es["hw_saved"]):
                _cap_sel(capture, "enc/hw%d" % l, pre[:, :Din] > 0)
        zc, std, kld = _new((B, Z + C), dev), _new((B, Z), dev), _new((1,), dev)
        ops.stamp("M mu|logvar done (waits for c now)")
        self.join(0)  # wake_sleep(): the discriminator step and c inference ran beside the encoder forward
        ops.stamp("M joined c")
        cvec = c_target.reshape(B, C).contiguous()
        zg = _new((B, 4 * Hg), dev)
        if fused_head:
            es["ctx"], es["mulv"] = _new((B, 2 * He), dev), _new((B, 2 * Z), dev)
            ops.code_head_fwd(es["sf"].c_final(), es["sr"].c_final(), w2f, self._fused("e2.b", P), eps, cvec,
                              Wih0[:, E:], es["ctx"], es["mulv"], zc, std, kld, zg)
        else:
            ops.reparam_kld_fwd(es["mulv"], eps, cvec, zc, std, kld)
        mulv = es["mulv"]
        # ---- generator (model/generator.py:45-64) --------------------------------------------------------------
        gen_emb = _new((NT, E), dev)
        ops.gather_rows(table, dec_word.reshape(-1), gen_emb,
                        mask=None if gen_drop_mask is None else gen_drop_mask.reshape(NT, E))
        # both layers in one wavefront launch when the shape allows (csrc/lstm.hip lstm2_*): layer 1 needs no xg then
        fuse2 = (cfg.decoder_num_layers == 2 and ops.lstm2_persistent_bytes(B, T, Hg, dev) > 0
                 and all(P["generator.rnn.%s_l%d" % (k, l)].is_contiguous() for k in ("weight_hh",) for l in (0, 1))
                 and P["generator.rnn.weight_ih_l1"].is_contiguous())
        g0, g1 = engine.LstmSeq(B, T, Hg, dev, False), engine.LstmSeq(B, T, Hg, dev, False, need_xg=not fuse2)
        ops.gemm(gen_emb, Wih0[:, :E].t(), g0.xg.view(NT, 4 * Hg), bias0=P["generator.rnn.bias_ih_l0"],
                 bias1=P["generator.rnn.bias_hh_l0"])
        if not fused_head:
            ops.gemm(zc, Wih0[:, E:].t(), zg)  # the tiled [z,c] input contributes once per sentence
        g0.xg_extra = zg
        g0.set_initial()
        g1.set_initial()
        ops.stamp("M generator input products done")
        self.exclusive()
        if fuse2:
            ops.lstm2_fwd(g0.xg, zg, P["generator.rnn.weight_hh_l0"], P["generator.rnn.weight_ih_l1"],
                          P["generator.rnn.weight_hh_l1"], P["generator.rnn.bias_ih_l1"], P["generator.rnn.bias_hh_l1"],
                          g0.hbuf, g0.cbuf, g0.gates, g1.hbuf, g1.cbuf, g1.gates, B, T, Hg)
            h0, h0ax = g0.hout_operand()
        else:
            engine.lstm_run_fwd([g0], [P["generator.rnn.weight_hh_l0"]])
            h0, h0ax = g0.hout_operand()
            ops.gemm(h0, P["generator.rnn.weight_ih_l1"].t(), g1.xg.view(NT, 4 * Hg), M=NT, K=Hg, a_m=h0ax, a_k=LIN,
                     bias0=P["generator.rnn.bias_ih_l1"], bias1=P["generator.rnn.bias_hh_l1"])
            engine.lstm_run_fwd([g1], [P["generator.rnn.weight_hh_l1"]])
        h1, h1ax = g1.hout_operand()
        ops.stamp("M generator LSTM fwd done")
        # bf16 mode: the logits-sized tensors (logits, dlogits and the transposed copy dW_fc reads) are HELD in bf16 when the
        # shapes allow 16-byte rows (ops.BF16_STORAGE; include/cgn_hip.h CGN_STORE_*): half the HBM bytes of the five
        # largest products' big operand. Everything else stays fp32.
        # (the bf16-stored A operand needs 16-byte loads on its partners too: word table [V,E], W_fc [V,H], h history)
        store16 = (ops.get_compute() == "bf16" and ops.BF16_STORAGE and V % 8 == 0 and NT % 8 == 0 and E % 4 == 0
                   and Hg % 4 == 0 and table.stride(0) % 4 == 0)
        logits = _new((NT, V), dev, torch.bfloat16 if store16 else torch.float32)
        # fused vocabulary projection + log-softmax + CE (generator.py:61 -> cgn.py:1086-1092): where the product runs the
        # X6 128-tile kernel its epilogue also leaves each row's per-chunk (max, exp-sum) of logits/tau, and a small kernel
        # combines them and reads the M target logits - the 107.5 MB of logits are not read back for the CE
        stats = None
        if ops.FUSED_LSE and not store16:
            _, stats = ops.gemm(h1, P["generator.fc.weight"].t(), logits, M=NT, K=Hg, a_m=h1ax, a_k=LIN,
                                bias0=P["generator.fc.bias"], lse_stats=inv_tau, dyn=self._dyn())
        else:
            ops.gemm(h1, P["generator.fc.weight"].t(), logits, M=NT, K=Hg, a_m=h1ax, a_k=LIN, bias0=P["generator.fc.bias"])
        # ---- temperature softmax + CE (cgn.py:1086-1092) -------------------------------------------------------
        tgt = target.reshape(-1)
        ops.stamp("M vocabulary projection done")
        if stats is not None:
            lse, nll, ce = ops.softmax_nll_from_stats(stats, logits, tgt, inv_tau, dyn=self._dyn())
        else:
            lse, nll, ce = ops.softmax_nll_fwd(logits, tgt, inv_tau, dyn=self._dyn())
        ops.stamp("M softmax/CE done")
        need_soft = use_disc or calc_z_loss
        # (a loss that this step does not compute reads as the engine's constant zero: no fill kernel on the critical path)
        c_loss = self._zero1
        z_loss = _new((1,), dev) if calc_z_loss else self._zero1
        F_disc = sum(self.Cos)
        # the c-reconstruction head and col2im + row dot product as single launches (csrc/fused.hip) where they apply
        fused_closs = ops.FUSED_CHAINS and use_disc and F_disc <= 1024 and B + 1 <= ops.FUSED_SCRATCH_WORDS
        rowdot = _new((NT,), dev)
        if need_soft:
            # ---- soft embedding (cgn.py:1103-1108): softmax(logits/tau) @ E, probabilities formed in the A loader
            out_emb = _new((NT, E), dev)
            ops.gemm(logits, table, out_emb, a_exp=(inv_tau, lse), dyn=self._dyn(), force_tile=SOFTEMB_TILE,
                     force_splitk=SOFTEMB_SPLITK)
            ops.stamp("M soft embedding done")
            d_out_emb = _new((NT, E), dev)
            if not (fused_closs and train):
                ops.fill(d_out_emb, 0.0)  # rows t = T-1 stay zero: out_embedding[:, :-1, :] (cgn.py:1108)
        if use_disc:
            # ---- c reconstruction through the frozen discriminator(s) (cgn.py:1114-1120) ---------------------------
            for a in range(C if fused_closs else 0):
                cw, cb = self._conv_params(P, a)
                fcw, fcb = self._fc_params(P, a)
                convs, Ps = engine.textcnn_conv_fwd(out_emb, B, L, E, T * E, cw, cb)  # rows t < L per sentence
                feat_d, arg_d = _new((B, F_disc), dev), _new((B, F_disc), dev, torch.int32)
                stats2 = _new((2,), dev)
                dconvs = ops.textcnn_head(convs, Ps, B, fcw, fcb, cvec[:, a], cfg.lambda_c, feat_d, arg_d, stats2)
                _cap_sel(capture, "soft%d/conv" % a, arg_d)
                c_loss = stats2[0:1] if a == 0 else ops.lincomb([(1.0, c_loss), (1.0, stats2[0:1])])
                if train:
                    dcols = ops.conv_dx_from_dconv(self.Ks, dconvs, [w.view(w.shape[0], -1) for w in cw], B, L, E)
                    ops.col2im_rowdot(self.Ks, dcols, B, L, T, E, d_out_emb, out_emb, rowdot, beta=0.0 if a == 0 else 1.0)
            for a in range(0 if fused_closs else C):
                cw, cb = self._conv_params(P, a)
                feat_d, arg_d = engine.textcnn_features_fwd(out_emb, B, L, E, T * E, cw, cb)  # rows t < L per sentence
                _cap_sel(capture, "soft%d/conv" % a, arg_d)
                fcw, fcb = self._fc_params(P, a)
                logits2, dlog2, stats2 = _new((B, 2), dev), _new((B, 2), dev), _new((2,), dev)
                y_c = _new((B,), dev, torch.int64)
                ops.float_to_index(cvec[:, a].contiguous() if C > 1 else cvec.reshape(B), y_c)
                ops.disc_head(feat_d, fcw, fcb, logits2, y=y_c, w_ce=cfg.lambda_c, dlogits=dlog2, stats=stats2)
                c_loss = stats2[0:1] if a == 0 else ops.lincomb([(1.0, c_loss), (1.0, stats2[0:1])])
                if train:
                    dfeat_d = _new(tuple(feat_d.shape), dev)
                    ops.disc_head_bwd(feat_d, fcw, dlog2, dfeat=dfeat_d)
                    tgt_dx = d_out_emb if a == 0 else _new((NT, E), dev)
                    if a > 0:
                        ops.fill(tgt_dx, 0.0)
                    ops.conv_bwd_dx_dense(self.Ks, self.Cos, ACT_RELU, 0, [w.view(w.shape[0], -1) for w in cw], dfeat_d,
                                    None, arg_d, B, L, E, tgt_dx, T * E)
                    if a > 0:
                        ops.add3d(tgt_dx.view(B, T, E), d_out_emb.view(B, T, E))
        if calc_z_loss:
            # ---- z reconstruction through the encoder (cgn.py:1122-1137) ------------------------------------------
            enc_in2 = ops.new2d(N, Din, dev)
            ops.copy3d(out_emb.view(B, T, E)[:, :L], enc_in2.unflatten(0, (B, L))[:, :, :E])
            ops.copy2d(tfeat, enc_in2[:, E:])
            es2 = self._encoder_fwd(enc_in2, B, L)
            if capture is not None:
                for l, (_, pre) in enumerate(es2["hw_saved"]):
                    _cap_sel(capture, "recon/hw%d" % l, pre[:, :Din] > 0)
            dmulv2 = _new((B, 2 * Z), dev)
            ops.mahalanobis(mulv, es2["mulv"], cfg.lambda_z, z_loss, dmulv2)
            if train:
                d_enc_in2 = self._encoder_bwd(es2, dmulv2, param_grads=False)
                # add the word-part gradient into d_out_emb rows (b, t < L)
                ops.add3d(d_enc_in2.unflatten(0, (B, L))[:, :, :E], d_out_emb.view(B, T, E)[:, :L])
        if self.dynamic:
            enc_loss = gen_loss = None  # kld_coef is a device scalar here: hip/graph.py forms the totals after replay
        else:
            enc_loss, gen_loss = self.total_losses(ce, kld, c_loss, z_loss, kld_coeff, use_disc, calc_z_loss)
        if not train:
            return ce, kld, z_loss, c_loss, enc_loss, gen_loss
        # ---- backward through softmax/CE + soft embedding, fused, in place over the logits ----------------------
        ops.stamp("M c-loss chain done")
        dlogits_t = _new((V, NT), dev, torch.bfloat16) if store16 else None  # k-contiguous A operand of dW_fc
        if need_soft:
            if not fused_closs or calc_z_loss:  # (the z-reconstruction pass added to d_out_emb after the fused dot product)
                ops.rowdot(d_out_emb, out_emb, rowdot)
            ops.gemm(d_out_emb, table.t(), logits, c_t=dlogits_t,
                     softmax_bwd=dict(logits=logits, lse=lse, rowdot=rowdot, target=tgt, inv_tau=inv_tau,
                                      ce_scale=79.0 * inv_tau / NT), dyn=self._dyn())
        else:
            # CE only (initial VAE training, cgn.py:639-644): same epilogue with an empty product (K = 0)
            ops.fill(rowdot, 0.0)
            # (explicit K = 0 over real operands: torch hands out a NULL data_ptr for zero-element views)
            ops.gemm(rowdot, table, logits, M=NT, N=V, K=0, a_m=Axis(1, 0, 0), a_k=LIN,
                     b_k=LIN, b_n=Axis(table.stride(0), 0, 0), c_t=dlogits_t,
                     softmax_bwd=dict(logits=logits, lse=lse, rowdot=rowdot, target=tgt, inv_tau=inv_tau,
                                      ce_scale=79.0 * inv_tau / NT), dyn=self._dyn())
        dlogits = logits
        ops.stamp("M dlogits done")
        dH1 = _new((NT, Hg), dev)
        ops.gemm(dlogits, P["generator.fc.weight"], dH1)
        ops.stamp("M dH1 done")
        # Parameter gradients of the generator: needed by nothing before its Adam step. With streams overlapped they are
        # queued (later) and forked behind the encoder's LSTM backward, the step's last persistent launch (see
        # exclusive()); otherwise they run here.
        self.begin_defer(capture)
        dl_rows = dlogits_t if store16 else dlogits.t()  # A[v, (b,t)] of dW_fc = dlogits^T . h1
        self.later(1, lambda: (ops.stamp("S1 deferred work begins (dW_fc)"),
                               ops.gemm(dl_rows, h1, G["generator.fc.weight"], N=Hg, K=NT, b_k=h1ax, b_n=LIN),
                               ops.stamp("S1 dW_fc done"),
                               ops.colsum(dlogits, G["generator.fc.bias"])))
        # ---- generator LSTM backward ---------------------------------------------------------------------------
        self.exclusive()
        if fuse2:
            # one wavefront launch: dg1_t feeds dh0_t = W_ih1^T dg1_t + W_hh0^T dg0_{t+1} inside the kernel
            dg1, dg0 = _new((B, T, 4 * Hg), dev), _new((B, T, 4 * Hg), dev)
            ops.lstm2_bwd(P["generator.rnn.weight_hh_l0"], P["generator.rnn.weight_ih_l1"],
                          P["generator.rnn.weight_hh_l1"], g0.cbuf, g0.gates, g1.cbuf, g1.gates, dH1, T * Hg, Hg,
                          dg0, dg1, B, T, Hg)
        else:
            dc1 = _new((B, Hg), dev)
            ops.fill(dc1, 0.0)
            (dg1,) = engine.lstm_run_bwd([g1], [P["generator.rnn.weight_hh_l1"]], [(dH1, T * Hg, Hg)], [None], [dc1])
            dH0 = _new((NT, Hg), dev)
            ops.gemm(dg1.view(NT, 4 * Hg), P["generator.rnn.weight_ih_l1"], dH0)
            dc0 = _new((B, Hg), dev)
            ops.fill(dc0, 0.0)
            self.exclusive()
            (dg0,) = engine.lstm_run_bwd([g0], [P["generator.rnn.weight_hh_l0"]], [(dH0, T * Hg, Hg)], [None], [dc0])
        group_gen = GROUP_TAIL and ops.get_compute() == "f32" and ops.GEMM_X6
        item1 = (g1, dg1, (h0, h0ax), G["generator.rnn.weight_ih_l1"], G["generator.rnn.weight_hh_l1"],
                 G["generator.rnn.bias_ih_l1"], G["generator.rnn.bias_hh_l1"])
        if not group_gen:
            self.later(1, lambda: engine.lstm_param_grads(*item1))
        ops.stamp("M generator LSTM bwd done")
        dWih0 = G["generator.rnn.weight_ih_l0"]
        dsum = _new((B, 4 * Hg), dev)
        dzc = _new((B, Z + C), dev)
        fused_head_bwd = fused_head and 4 * Hg <= 4096
        if fused_head_bwd:
            # sum over time, d[z,c], reparametrisation backward (+ KLD term) and d(context) in one launch (csrc/fused.hip)
            dmulv = _new((B, 2 * Z), dev)
            dc_enc = (_new((B, He), dev), _new((B, He), dev))
            ops.code_head_bwd(dg0, Wih0[:, E:], mulv, eps, w2f, float(kld_coeff) / B, dsum, dzc, dmulv, dc_enc[0],
                              dc_enc[1], dyn=self._dyn(2))
        else:
            ops.sum_time(dg0, dsum)
        item0 = (g0, dg0, (gen_emb, Axis(E, 0, 0)), dWih0[:, :E], G["generator.rnn.weight_hh_l0"],
                 G["generator.rnn.bias_ih_l0"], G["generator.rnn.bias_hh_l0"])
        if group_gen:  # both layers' weight-gradient products over the history buffers as one launch
            self.later(1, lambda: (engine.lstm_param_grads_multi([item1, item0]), ops.gemm(dsum.t(), zc, dWih0[:, E:])))
        else:
            self.later(1, lambda: (engine.lstm_param_grads(*item0), ops.gemm(dsum.t(), zc, dWih0[:, E:])))
        if not fused_head_bwd:
            ops.gemm(dsum, Wih0[:, E:], dzc)  # g = dzc[:, :Z] = d(79*CE + recon)/dz  (cgn.py:1099, 1142)
        if capture is not None:
            self.join(1)
            capture["g"] = dzc[:, :Z].clone()
            capture["gen_grads"] = {n: g.clone() for n, g in self.arenas["gen"].grads.items()}
        if not joint:
            # the generator's Adam step follows its gradients (cgn.py:1143); the encoder backward does not wait for it.
            # dzc above was the last read of a generator weight on the main stream. (Data parallel: _optim_step starts
            # the all-reduce of the generator's gradient arena here, on the stream that formed the gradients, and waits
            # for it on that stream: it travels beside the encoder's backward pass.)
            self.later(1, lambda: (ops.stamp("S1 generator gradients done"), self._optim_step(opts[1]),
                                   ops.stamp("S1 generator Adam done")))
        # ---- encoder backward (cgn.py:1145-1149): kld_coef*KLD + z_out.backward(g) ---------------------------------
        self._split_enc = bool(self.dp and not joint and self._enc_cut is not None)
        if fused_head_bwd:
            d_enc_in = self._encoder_bwd(es, dmulv, param_grads=True, dc=dc_enc)
        else:
            dmulv = _new((B, 2 * Z), dev)
            ops.reparam_kld_bwd(mulv, eps, dzc, float(kld_coeff) / B, dmulv, dyn=self._dyn(2))
            d_enc_in = self._encoder_bwd(es, dmulv, param_grads=True)
        ops.stamp("M highway bwd (dX) done")
        dtf = _new((N, S), dev)
        ops.copy2d(d_enc_in[:, E:], dtf)
        engine.tdnn_bwd(tsaved, tk, dtf, self._tdnn_kernels(G), G["embedding.char_embed.weight"])
        ops.stamp("M TDNN + char embedding bwd done")
        ops.PLAN_HINT = None
        self.flush()  # (nothing is left unless the encoder backward ran without parameter gradients)
        self.join(0, 1)
        ops.stamp("M joined side streams")
        if capture is not None:
            capture["enc_grads"] = {n: g.clone() for n, g in self.arenas["enc"].grads.items()}
        if self.dp and joint:  # initial VAE training: one optimizer over everything (cgn.py:623,645)
            self._comm([("begin", "enc"), ("begin", "gen"), ("wait", "enc")])
            self._adam(opts[0])
            self._comm([("wait", "gen")])
            self._adam(opts[1])
        else:
            self._optim_step(opts[0])  # cgn.py:1149
            if joint:
                self._optim_step(opts[1])
        return ce, kld, z_loss, c_loss, enc_loss, gen_loss
