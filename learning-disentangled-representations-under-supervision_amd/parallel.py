"""Data parallelism for the CGN step: one process per GPU, replicated weights, per-rank batch shard, gradient
sum all-reduce (RCCL over xGMI on MI355X; gloo on CPU for tests) of each optimizer group's flat gradient arena right
before that group's Adam step. The 1/world_size scaling is folded into the Adam kernel's gradient read.

The reference has no distributed code; every loss of the step is a mean over independent sentences
(§8e), so averaging per-rank gradients of equal-size shards equals one step over the concatenated batch.

Buckets: xGMI is point-to-point, so a ring all-reduce is bound per link; a few large contiguous messages are used
(BUCKET_BYTES) instead of one call per tensor. A group's buckets are started (begin) as soon as its gradients are final
and waited for (wait) right before its Adam kernel: the generator's 57 MB travel while the encoder's backward pass
runs, the encoder's while the generator's Adam kernel runs (hip/step.py, hip/graph.py).
"""
import torch
import torch.distributed as dist

BUCKET_BYTES = 32 << 20


class BucketedAllReduce:
    def __init__(self, group=None, bucket_bytes=BUCKET_BYTES, force=False):
        """force: issue the collectives even in a one-rank group (single-GPU rehearsal of the data-parallel schedule)."""
        self.group = group
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.force = force

    def _active(self):
        return dist.is_initialized() and (self.force or dist.get_world_size(self.group) > 1)

    def _host_staged(self, flat):
        """True when the collective moves a device tensor through the host (gloo; rehearsals on one GPU only).
        ProcessGroupGloo stages HIP tensors on a HIGH-PRIORITY side stream that waits for an event on the caller's
        stream. On this ROCm stack a high-priority stream that waits behind queued work of a normal-priority stream
        (a replayed hipGraph segment) stalls that work for seconds to indefinitely - measured without any process group
        by tools/hp_stream_probe.py (profiles/r02_dp2_hang_probe.txt: 4 ms behind a normal-priority side stream, > 60 s
        behind a high-priority one). This was round 1's "segmented-graph replay hangs under gloo". RCCL's side stream
        has normal priority (never request is_high_priority_stream), so the RCCL path has no such wait."""
        if not flat.is_cuda:
            return False
        try:
            return "gloo" in str(dist.get_backend(self.group)).lower() and "nccl" not in str(
                dist.get_backend(self.group)).lower()
        except Exception:  # noqa: BLE001
            return False

    def begin(self, flat):
        """Start the in-place sum of `flat` (1-D fp32 tensor) over all ranks; returns the handles for wait().
        On HIP the collectives run on RCCL's own stream, ordered after everything issued so far on the current
        stream, so kernels launched between begin() and wait() overlap the transfer."""
        if not self._active():
            return []
        if self._host_staged(flat):
            # gloo blocks the host in wait() anyway: let the caller's stream drain first, so that gloo's high-priority
            # stream finds its event already complete and never waits behind queued work (see _host_staged)
            torch.cuda.current_stream(flat.device).synchronize()
        n = flat.numel()
        handles = []
        # RCCL (device tensors): SYNCHRONOUS collectives. torch (>= 2.8) enqueues a synchronous NCCL collective on the
        # CALLER's current stream - here the stream that formed the gradients - instead of the process group's own
        # stream. That is the point: HIP multiplexes streams onto 4 hardware queues, this engine already uses three
        # (main + two side streams), and a fifth stream whose first packet waits for a side stream's event blocks
        # whatever shares its hardware queue (measured: the whole main chain, 0.5 ms per step). In-stream, the
        # all-reduce is ordered like any other kernel of that stream and needs no event at all; "wait" is a no-op.
        sync = flat.is_cuda and not self._host_staged(flat)
        for lo in range(0, n, self.bucket_elems):
            hi = min(n, lo + self.bucket_elems)
            h = dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=not sync)
            if not sync:
                handles.append(h)
        return handles

    @staticmethod
    def wait(handles):
        for h in handles:
            h.wait()  # on HIP: makes the current stream wait for the collective (no host block)

    def __call__(self, flat):
        """Sum `flat` in place over all ranks (begin + wait)."""
        self.wait(self.begin(flat))


def quiesce_before_capture(device=None):
    """Call before ANY hipGraph capture in a process that has an RCCL process group. ProcessGroupNCCL's watchdog thread
    polls the end event of every collective it has not yet seen complete (hipEventQuery, every 100 ms); on ROCm 7.2 such
    a query fails with hipErrorCapturedEvent while the stream the event was last recorded on is being captured - even for
    an event recorded long before the capture began - and the watchdog then terminates the process (seen once in ~10 runs
    of tests/test_graph_gpu.py: the eager warm-up step's collectives were still on the watchdog's list when the capture
    of the same streams started). Draining the device and giving the watchdog a few polls empties its list."""
    if dist.is_available() and dist.is_initialized() and torch.cuda.is_available():
        import time
        torch.cuda.synchronize(device)
        time.sleep(0.5)


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun contract).
    Returns (rank, world_size, local_rank)."""
    import os
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" is RCCL on ROCm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_batch(tensors, rank, world):
    """Rank r takes sentences [r*B/world, (r+1)*B/world) of every tensor whose first dim is the global batch."""
    out = {}
    for k, v in tensors.items():
        B = v.shape[0]
        if B % world:
            raise ValueError("global batch %d not divisible by world size %d" % (B, world))
        per = B // world
        out[k] = v[rank * per:(rank + 1) * per].contiguous()
    return out


def attach(model, group=None, force=False):
    """Make `model`'s fused step data-parallel over the initialised process group. force=True keeps the data-parallel
    schedule (and its collectives) on even when the group has one rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    model.set_data_parallel(BucketedAllReduce(group, force=force), world, force=force)
    return world
