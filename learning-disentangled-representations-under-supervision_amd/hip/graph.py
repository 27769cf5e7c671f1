"""The wake-sleep step as a replayable hipGraph.

Why: the step is ~350 kernel launches of 3-400 us each; issued one by one from Python the host needs as long to launch
them as the MI355X needs to run them. Captured once into a hipGraph the whole step is one launch, and because the
graph carries the dependency edges of the side streams (CGNStep.side/join), the latency-bound LSTM recurrences and the
parameter-gradient GEMMs overlap with the dX chain.

What changes from call to call lives in device memory, not in kernel arguments:
  * the batch: static input tensors, refreshed with copy_() before a replay;
  * the schedules (1/tau, CE scale, kld_coef/B): CGNStep.dyn, refreshed by one small async H2D copy;
  * Adam's step counts: AdamState.step_dev, advanced by a node of the graph itself.
Nothing of the reference's semantics changes: the same kernels run in the same order on the same data as the eager
step (tests/test_graph_gpu.py checks replayed training against eager training: losses 1e-6, weights 2e-6 — Adam's
bias corrections are formed in double on the device instead of on the host).

Data parallel runs (world_size > 1) replay the step as a CHAIN of graph segments cut at the communication points of
CGNStep's data-parallel schedule (step.py `_comm`): segment, RCCL all-reduce calls issued eagerly on RCCL's own stream,
next segment. The collectives themselves are never captured (nothing here depends on RCCL being capturable); what the
segmentation keeps is the one-launch-per-segment host cost and the overlap: the generator's gradient all-reduce
travels while the encoder-backward segment runs, the encoder's while the generator's Adam segment runs.
"""
import torch

from . import ops

INPUT_KEYS = ("x_labelled", "y_labelled", "x_generated", "c_generated", "mask_labelled", "mask_generated", "enc_word",
              "enc_char", "dec_word", "target", "c_uniform", "eps")


class _SegmentRecorder:
    """Captures one pass of the step into consecutive hipGraphs that share a memory pool; CGNStep._comm calls cut()."""

    def __init__(self, thread_local):
        self.segments = []  # [(graph, [actions run after it])]
        self.pool = torch.cuda.graph_pool_handle()
        self.g = None
        # with a process group alive, its watchdog thread may touch HIP events while this thread captures
        self.kw = dict(capture_error_mode="thread_local") if thread_local else {}

    def begin(self):
        self.g = torch.cuda.CUDAGraph()
        self.g.capture_begin(pool=self.pool, **self.kw)

    def cut(self, actions):
        self.g.capture_end()
        self.segments.append((self.g, actions))
        self.begin()

    def end(self):
        self.g.capture_end()
        self.segments.append((self.g, []))
        self.g = None

    def abort(self):
        if self.g is not None:
            try:
                self.g.capture_end()
            except Exception:
                pass
            self.g = None


class GraphedWakeSleep:
    def __init__(self, eng, opts, data, calc_z_loss=False, overlap=True, warmup_kld=0.5, warmup_tau=1.0):
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
            ops.KEEP = []
            try:
                if eng.dp:
                    import gc
                    gc.collect()
                    torch.cuda.empty_cache()
                    rec = _SegmentRecorder(thread_local=torch.distributed.is_initialized())
                    eng.recorder = rec
                    try:
                        with torch.cuda.stream(self.stream):
                            rec.begin()
                            try:
                                self.out = self._run()
                                rec.end()
                            except BaseException:
                                rec.abort()
                                raise
                    finally:
                        eng.recorder = None
                    self.segments = rec.segments
                else:
                    with torch.cuda.graph(self.graph, stream=self.stream):
                        self.out = self._run()
                    self.segments = [(self.graph, [])]
            finally:
                self.keep, ops.KEEP = ops.KEEP, None  # temporaries of the captured step stay alive with the graph
        finally:
            eng.overlap, eng.dynamic = prev

    def _slot(self):
        self.calls += 1
        return self.pinned[self.calls % self.pinned.shape[0]]

    def _run(self):
        return self.eng.wake_sleep(self.opts, self.static, 0.0, 1.0, calc_z_loss=self.calc_z)

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
        eng.set_step_scalars(kld_coeff, temperature, self.NT, self.B, self._slot())
        for i, (g, actions) in enumerate(self.segments):
            self.progress = (self.calls, i, "replay")
            g.replay()
            for kind, group in actions:
                self.progress = (self.calls, i, kind + " " + group)
                eng.do_comm(kind, group)
        self.progress = (self.calls, len(self.segments), "done")
        self._bump_host_steps()
        d_losses, c, eg = self.out
        ce, kld, z_loss, c_loss = eg[:4]
        if totals:
            enc, gen = eng.total_losses(ce, kld, c_loss, z_loss, kld_coeff, True, self.calc_z)
        else:
            enc = gen = None
        return d_losses, c, (ce, kld, z_loss, c_loss, enc, gen)
