"""The wake-sleep step as a replayable hipGraph.

Why: the step is ~140 kernel launches of 3-270 us each; issued one by one from Python the host needs longer to launch
them than the MI355X needs to run them. The step is captured once and replayed as ONE hipGraph PER BRANCH
(_BranchRecorder): the main chain is cut wherever CGNStep.side() forks or join() joins, every side region is a graph of
its own, and a replay launches them in recorded order on the caller's stream and two explicit side streams with events
for fork and join - so the discriminator step runs beside the encoder forward and the parameter-gradient GEMMs beside
the dX chain by the program's decision, not the runtime's. (Round 1 captured one graph with internal branches; ROCm
mostly serialised them and once mis-ordered a three-successor node.) tests/test_graph_fullsize_gpu.py holds this replay
to the eager step at the benchmark's sizes.

What changes from call to call lives in device memory, not in kernel arguments:
  * the batch: static input tensors, refreshed with copy_() before a replay;
  * the schedules (1/tau, CE scale, kld_coef/B): CGNStep.dyn, refreshed by one small async H2D copy;
  * Adam's step counts: AdamState.step_dev, advanced by a node of the graph itself.
Nothing of the reference's semantics changes: the same kernels run in the same order on the same data as the eager
step (tests/test_graph_gpu.py checks replayed training against eager training: losses 1e-6, weights 2e-6 — Adam's
bias corrections are formed in double on the device instead of on the host).

Data parallel runs (world_size > 1) replay the SAME per-branch graphs; every communication point of the step (CGNStep._comm:
begin / wait of one optimizer group's gradient all-reduce) cuts the graph of the stream it occurs on - the side stream that
formed the group's last gradient, or the main stream - and the RCCL calls are issued eagerly between the two halves with
that stream current. The collectives themselves are never captured (nothing here depends on RCCL being capturable) and no
other stream is joined for them: the generator's all-reduce travels beside the encoder's dX chain, the discriminator's
beside the encoder forward (round 2 cut ONE chain of segments at every collective and gave up the deferral: +13 % on one
rank).
"""
import warnings

import torch

from . import ops

INPUT_KEYS = ("x_labelled", "y_labelled", "x_generated", "c_generated", "mask_labelled", "mask_generated", "enc_word",
              "enc_char", "dec_word", "target", "c_uniform", "eps")


class _BranchRecorder:
    """Captures one pass of the step as ONE hipGraph PER BRANCH: the main chain is cut wherever a side region begins or
    a side stream is joined, and every side region (CGNStep.side) becomes a graph of its own. A replay launches the
    graphs in recorded order, main graphs on the caller's stream, side graphs on explicit side streams, with events for
    the fork ("behind everything the main stream was given so far") and the join.

    Why not one graph with internal branches (round 1): which hardware queue ROCm gives a graph-internal branch is not
    under the program's control, and with HIP's 4 hardware queues per process two branches regularly shared one and
    simply ran one after the other (profiles/r02: the discriminator step never overlapped the encoder forward, the
    generator's parameter gradients never overlapped the encoder's dX chain). Explicit streams make the overlap the
    program's decision; the graphs share one memory pool and every temporary of the recorded step is kept alive
    (ops.KEEP), so graphs that run side by side never alias memory."""

    parallel = True

    def __init__(self, main_stream, thread_local):
        self.main_stream = main_stream
        self.pool = torch.cuda.graph_pool_handle()
        self.kw = dict(capture_error_mode="thread_local") if thread_local else {}
        # ("main", graph, [side streams to wait for first]) | ("side", graph, k, forks) | ("comm", actions, k | None):
        # forks = True for the graph that opens a side region (it starts behind the fork event), False for the graph that
        # continues the region behind a communication point (same stream: stream order)
        self.segments = []
        self.g = None        # main graph being captured
        self.waits = []      # joins that the NEXT main graph starts with
        self.side_g = None
        self.side_k = None
        self.side_first = True

    def _begin_main(self):
        self.g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(self.main_stream):
            self.g.capture_begin(pool=self.pool, **self.kw)

    def _end_main(self):
        if self.g is not None:
            with torch.cuda.stream(self.main_stream), warnings.catch_warnings():
                # a join right behind a fork leaves nothing between them: torch warns about the empty graph
                warnings.filterwarnings("ignore", message="The CUDA Graph is empty")
                self.g.capture_end()
            self.segments.append(("main", self.g, self.waits))
            self.g, self.waits = None, []

    def begin(self):
        self._begin_main()

    def _begin_side_graph(self):
        self.side_g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(self._side_stream):
            self.side_g.capture_begin(pool=self.pool, **self.kw)

    def _end_side_graph(self):
        with torch.cuda.stream(self._side_stream), warnings.catch_warnings():
            warnings.filterwarnings("ignore", message="The CUDA Graph is empty")
            self.side_g.capture_end()
        self.segments.append(("side", self.side_g, self.side_k, self.side_first))
        self.side_g, self.side_first = None, False

    def begin_side(self, k, stream):
        self._end_main()
        self._side_stream, self.side_k, self.side_first = stream, k, True
        self._begin_side_graph()

    def end_side(self, k):
        self._end_side_graph()
        self.side_k = None
        self._begin_main()

    def comm(self, actions):
        """A communication point on the stream being captured right now: its graph ends, the actions follow it, a new
        graph continues the stream."""
        if self.side_g is not None:
            self._end_side_graph()
            self.segments.append(("comm", actions, self.side_k))
            self._begin_side_graph()
        else:
            self._end_main()
            self.segments.append(("comm", actions, None))
            self._begin_main()

    def join(self, k):
        self._end_main()
        self.waits.append(k)
        self._begin_main()

    def end(self):
        self._end_main()

    def abort(self):
        for g, st in ((self.side_g, getattr(self, "_side_stream", None)), (self.g, self.main_stream)):
            if g is not None:
                try:
                    with torch.cuda.stream(st):
                        g.capture_end()
                except Exception:
                    pass
        self.g = self.side_g = None


class GraphedWakeSleep:
    def __init__(self, eng, opts, data, calc_z_loss=False, overlap=True, warmup_kld=0.5, warmup_tau=1.0,
                 branch_graphs=True):
        """eng: CGNStep; opts: dict(disc, enc, gen) of AdamState; data: dict of device tensors giving the SHAPES of
        every later batch (their current contents are used for ONE real warm-up step, which trains the model like any
        other step)."""
        self.eng, self.opts, self.calc_z = eng, opts, calc_z_loss
        self.static = {k: data[k].clone() for k in INPUT_KEYS if data.get(k) is not None}
        self.B = self.static["enc_word"].shape[0]
        self.NT = self.static["dec_word"].numel()
        # ring of pinned scalar blocks: the async H2D copy of step n may still be queued when the host prepares n+1
        self.pinned = torch.zeros(256, 4, dtype=torch.float32).pin_memory()
        self.calls = 0
        self._sticky_host = None
        self.branches = None  # per-branch graphs (single-GPU overlapped replay), see _BranchRecorder
        self.stream = torch.cuda.Stream(eng.dev)
        self.graph = torch.cuda.CUDAGraph()
        for st in opts.values():
            st.sync_device_step()
        prev = (eng.overlap, eng.dynamic)
        eng.overlap, eng.dynamic = overlap, True
        try:
            self.stream.wait_stream(torch.cuda.current_stream(eng.dev))
            # The warm-up is a REAL training step issued over the capture stream and its side streams: its temporaries
            # are read across streams, so they are kept alive (ops.KEEP) until the device has drained, exactly like the
            # captured step's (torch's per-stream caching allocator would otherwise recycle a block under a side-stream
            # reader).
            ops.KEEP = []
            try:
                with torch.cuda.stream(self.stream):
                    # warm-up on the capture stream and its side streams: creates the per-stream workspaces and loads
                    # every kernel before capture (allocation and module loading are illegal while capturing)
                    eng.set_step_scalars(warmup_kld, warmup_tau, self.NT, self.B, self._slot())
                    self._run()
                    self._bump_host_steps()
                torch.cuda.current_stream(eng.dev).wait_stream(self.stream)
                torch.cuda.synchronize(eng.dev)
            finally:
                ops.KEEP = None
            from .. import parallel
            parallel.quiesce_before_capture(eng.dev)  # (RCCL watchdog vs. capture: see there)
            ops.KEEP = []
            try:
                with ops.capture_without_gc():  # a collector run inside a capture can abort the process (ops.py)
                    if eng.dp or (overlap and branch_graphs):
                        torch.cuda.empty_cache()
                        rec = _BranchRecorder(self.stream, thread_local=torch.distributed.is_initialized())
                        eng.recorder = rec
                        try:
                            rec.begin()
                            try:
                                with torch.cuda.stream(self.stream):
                                    self.out = self._run()
                                rec.end()
                            except BaseException:
                                rec.abort()
                                raise
                        finally:
                            eng.recorder = None
                        self.branches = rec.segments
                        self.segments = None
                    else:
                        with torch.cuda.graph(self.graph, stream=self.stream):
                            self.out = self._run()
                        self.segments = [(self.graph, [])]
            finally:
                self.keep, ops.KEEP = ops.KEEP, None  # temporaries of the captured step stay alive with the graph
        finally:
            eng.overlap, eng.dynamic = prev

    def _replay_branches(self):
        """Main graphs on the current stream, side graphs on the engine's side streams; a side graph starts behind
        everything the main stream has been given so far, a main graph that follows a join waits for that side stream.
        The host hands the main graph that FOLLOWS a fork to the GPU before the side graphs of that fork (the fork
        event is recorded where the step forks, so the order on the device is the recorded one): the main chain is the
        critical path and should never wait for the host to finish launching side work."""
        eng = self.eng
        main = torch.cuda.current_stream(eng.dev)
        sides = eng._side_streams
        done, held = {}, []

        def launch_sides():
            for item in held:
                k = item[2]
                st = sides[k]
                with torch.cuda.stream(st):
                    if item[0] == "side":
                        if item[3] is not None:
                            st.wait_event(item[3])
                        item[1].replay()
                    else:  # communication point on this side stream (RCCL orders itself behind the stream's work so far)
                        for kind, group in item[1]:
                            self.progress = (self.calls, "S%d" % k, kind + " " + group)
                            eng.do_comm(kind, group)
                    ev = torch.cuda.Event()
                    ev.record(st)
                done[k] = ev
            del held[:]

        for seg in self.branches:
            kind = seg[0]
            if kind == "main":
                g, waits = seg[1], seg[2]
                if waits:
                    launch_sides()
                    for k in waits:
                        if k in done:
                            main.wait_event(done.pop(k))
                g.replay()
                launch_sides()
            elif kind == "side":
                fork = None
                if seg[3]:  # opens a side region: behind everything the main stream has been given so far
                    fork = torch.cuda.Event()
                    fork.record(main)
                held.append(("side", seg[1], seg[2], fork))
            elif seg[2] is not None:  # communication point on a side stream: keeps its place in that stream's sequence
                held.append(("comm", seg[1], seg[2]))
            else:  # communication point on the main stream
                launch_sides()
                for kind_, group in seg[1]:
                    self.progress = (self.calls, "M", kind_ + " " + group)
                    eng.do_comm(kind_, group)
        launch_sides()
        for ev in done.values():  # nothing of this step may still run when the caller's stream moves on
            main.wait_event(ev)

    def _slot(self):
        self.calls += 1
        return self.pinned[self.calls % self.pinned.shape[0]]

    def _run(self):
        return self.eng.wake_sleep(self.opts, self.static, 0.0, 1.0, calc_z_loss=self.calc_z)

    # A persistent LSTM launch whose bounded wait ran out leaves a sticky device word (include/cgn_hip.h); every replay
    # after it trains on invalid activations. A caller that never synchronises (no logging) would learn of it only at the
    # end of its loop, so every STICKY_EVERY-th replay copies the words to pinned host memory behind itself (async, no
    # synchronisation) and every call looks at what earlier copies brought back: the loop stops at most
    # 2 * STICKY_EVERY replays after the launch that timed out.
    STICKY_EVERY = 8

    def _watch_sticky(self):
        if self.calls % self.STICKY_EVERY:
            return
        dev = self.eng.dev
        didx = dev.index if dev.index is not None else torch.cuda.current_device()
        bufs = [ws for (idx, _), ws in ops._lstm_sync.items() if idx == didx]
        if not bufs:
            return
        if self._sticky_host is None or self._sticky_host.numel() != len(bufs):
            self._sticky_host = torch.zeros(len(bufs), dtype=torch.int32).pin_memory()
        for i, ws in enumerate(bufs):
            self._sticky_host[i:i + 1].copy_(ws[-32:-31], non_blocking=True)

    def _check_sticky(self):
        h = self._sticky_host
        if h is not None and bool(h.any()):
            h.zero_()  # (the next watch copy brings the device word back if it is still set)
            raise ops.LstmTimeoutError(
                "a persistent LSTM launch ran out of a bounded wait some replays ago (sticky flag seen by the replay "
                "loop's asynchronous watch): activations since then are invalid. Reload the last checkpoint and rerun "
                "with CGN_LSTM_PERSISTENT=0 (ops.clear_lstm_sticky_timeouts() re-arms the persistent kernels).")

    def _bump_host_steps(self):
        for st in self.opts.values():
            st.step += 1

    def load(self, data):
        """Copy a new batch into the static input tensors (device-to-device or pinned-host-to-device, async)."""
        for k, t in self.static.items():
            src = data[k]
            if src.shape != t.shape:
                raise ValueError("%s: graph was captured for shape %s, got %s" % (k, tuple(t.shape), tuple(src.shape)))
            t.copy_(src, non_blocking=True)

    def __call__(self, kld_coeff, temperature, data=None, totals=True):
        """Replay one step on the current stream. Returns (disc 5-tuple, c, (ce, kld, z_loss, c_loss, enc, gen))."""
        eng = self.eng
        if data is not None:
            self.load(data)
        # The replay runs on this object's OWN stream (the caller's stream waits for it at the end). Why: HIP multiplexes
        # streams onto 4 hardware queues in creation order; the caller's stream is normally the null stream (queue 0), and
        # the 4th stream created after it - RCCL's internal stream in a data-parallel run - lands on queue 0 as well. A
        # collective that waits for a side stream's gradients then blocks the queue's head, and with it the whole main
        # chain (measured: the dX chain started 620 us late, --force-dp 3.38 ms against 2.98 ms; on its own stream
        # [queue 1, idle otherwise] the main chain shares a queue with nothing).
        self._check_sticky()
        cur = torch.cuda.current_stream(eng.dev)
        own = cur.cuda_stream != self.stream.cuda_stream
        if own:
            self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            eng.set_step_scalars(kld_coeff, temperature, self.NT, self.B, self._slot())
            if self.branches is not None:
                self._replay_branches()
            for i, (g, actions) in enumerate(self.segments or ()):  # (one graph: overlap off, or branch_graphs=False)
                self.progress = (self.calls, i, "replay")
                g.replay()
            self._watch_sticky()
        if own:
            cur.wait_stream(self.stream)
        self.progress = (self.calls, "-", "done")
        self._bump_host_steps()
        d_losses, c, eg = self.out
        ce, kld, z_loss, c_loss = eg[:4]
        if totals:
            enc, gen = eng.total_losses(ce, kld, c_loss, z_loss, kld_coeff, True, self.calc_z)
        else:
            enc = gen = None
        return d_losses, c, (ce, kld, z_loss, c_loss, enc, gen)
