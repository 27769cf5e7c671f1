"""Flat fp32 parameter arenas: one contiguous buffer per optimizer group for params, grads and Adam moments.

Why: (1) Adam for a whole group is ONE streaming kernel (cgn_adam_f32) instead of a launch per tensor; (2) the
data-parallel gradient all-reduce is a handful of large contiguous buckets per group (RCCL over xGMI is per-link
bound, so few large messages beat many small ones); (3) `zero_grad` is one fill. Module parameters become views into
the arena (same names/shapes, so state_dict round-trips with the reference's checkpoints).
"""
import torch

from . import ops

ALIGN = 64  # floats (256 B): keeps every tensor 16-byte aligned for float4 access


class ParamArena:
    def __init__(self, named_params, device):
        """named_params: list of (name, nn.Parameter). Parameters are re-pointed at views of the arena."""
        self.names = [n for n, _ in named_params]
        self.offsets, off = {}, 0
        for n, p in named_params:
            shape = tuple(p.shape)
            # big matrices whose rows are not a multiple of 4 floats (e.g. [*, 825] highway / LSTM input weights,
            # [*, 365] generator input weights) get a padded row stride so that GEMM operand rows are 16-byte aligned
            ld = shape[-1]
            if p.dim() == 2 and shape[0] >= 64 and shape[1] >= 64 and shape[1] % 4:
                ld = (shape[1] + 3) // 4 * 4
            ne = p.numel() // shape[-1] * ld if p.dim() >= 1 and shape[-1] else p.numel()
            self.offsets[n] = (off, ne, shape, ld)
            off += (ne + ALIGN - 1) // ALIGN * ALIGN
        self.size = off
        self.p = torch.zeros(off, dtype=torch.float32, device=device)
        self.g = torch.zeros(off, dtype=torch.float32, device=device)
        self.params, self.grads = {}, {}
        for n, p in named_params:
            o, ne, shape, ld = self.offsets[n]

            def carve(flat):
                if len(shape) == 2 and ld != shape[1]:
                    return flat[o:o + ne].view(shape[0], ld)[:, :shape[1]]
                return flat[o:o + ne].view(shape)

            view = carve(self.p)
            view.copy_(p.data)
            p.data = view
            gview = carve(self.g)
            p.grad = gview
            self.params[n] = p
            self.grads[n] = gview

    def owns(self, named_params):
        """True while every parameter still aliases this arena (a .cuda()/.to() on the module breaks that)."""
        base = self.p.untyped_storage().data_ptr()
        return all(p.data.untyped_storage().data_ptr() == base for _, p in named_params)

    def zero_grad(self):
        if self.size:
            ops.fill(self.g, 0.0)

    def new_adam(self):
        """A fresh optimizer state for this group (the reference creates a new torch Adam per step-closure factory)."""
        return AdamState(self)


class AdamState:
    def __init__(self, arena):
        self.arena = arena
        self.m = torch.zeros(arena.size, dtype=torch.float32, device=arena.p.device)
        self.v = torch.zeros(arena.size, dtype=torch.float32, device=arena.p.device)
        self.step = 0
        # device-resident copy of the step count, used when the step is replayed from a captured hipGraph (the graph
        # bumps it with cgn_incr_i32 and the Adam kernel forms the bias corrections from it)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=arena.p.device)

    def apply(self, lr, grad_scale=1.0, betas=(0.9, 0.999), eps=1e-8, dynamic=False):
        """One Adam update. dynamic=True reads (and first increments) the device step count instead of the host's;
        the caller (hip/graph.py) keeps `self.step` in line with it."""
        a = self.arena
        if dynamic:
            ops.incr(self.step_dev)
            if a.size:
                ops.adam(a.p, a.g, self.m, self.v, lr, 0, betas[0], betas[1], eps, grad_scale, dyn_step=self.step_dev)
            return
        self.step += 1
        if a.size:
            ops.adam(a.p, a.g, self.m, self.v, lr, self.step, betas[0], betas[1], eps, grad_scale)

    def sync_device_step(self):
        self.step_dev.fill_(self.step)
