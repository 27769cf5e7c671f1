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

Data parallel runs (world_size > 1) stay eager this round: the bucketed RCCL all-reduce sits between graph segments
and capturing it has not been validated on a multi-GPU node yet.
"""
import torch

from . import ops

INPUT_KEYS = ("x_labelled", "y_labelled", "x_generated", "c_generated", "mask_labelled", "mask_generated", "enc_word",
              "enc_char", "dec_word", "target", "c_uniform", "eps")


class GraphedWakeSleep:
    def __init__(self, eng, opts, data, calc_z_loss=False, overlap=True, warmup_kld=0.5, warmup_tau=1.0):
        """eng: CGNStep; opts: dict(disc, enc, gen) of AdamState; data: dict of device tensors giving the SHAPES of
        every later batch (their current contents are used for ONE real warm-up step, which trains the model like any
        other step)."""
        if eng.world_size > 1:
            raise RuntimeError("graph replay of the data-parallel step is not enabled: run the eager step")
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
            with torch.cuda.stream(self.stream):
                # warm-up on the capture stream and its side streams: creates the per-stream workspaces and loads every
                # kernel before capture (allocation and module loading are illegal while capturing)
                eng.set_step_scalars(warmup_kld, warmup_tau, self.NT, self.B, self._slot())
                self._run()
                self._bump_host_steps()
            torch.cuda.current_stream(eng.dev).wait_stream(self.stream)
            torch.cuda.synchronize(eng.dev)
            ops.KEEP = []
            try:
                with torch.cuda.graph(self.graph, stream=self.stream):
                    self.out = self._run()
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
        self.graph.replay()
        self._bump_host_steps()
        d_losses, c, eg = self.out
        ce, kld, z_loss, c_loss = eg[:4]
        if totals:
            enc, gen = eng.total_losses(ce, kld, c_loss, z_loss, kld_coeff, True, self.calc_z)
        else:
            enc = gen = None
        return d_losses, c, (ce, kld, z_loss, c_loss, enc, gen)
