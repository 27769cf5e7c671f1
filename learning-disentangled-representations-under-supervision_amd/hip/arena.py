"""Flat fp32 parameter arenas: one contiguous buffer per optimizer group for params, grads and Adam moments.

Why: (1) Adam for a whole group is ONE streaming kernel (cgn_adam_f32) instead of a launch per tensor; (2) the
data-parallel gradient all-reduce is a handful of large contiguous buckets per group (RCCL over xGMI is per-link
bound, so few large messages beat many small ones); (3) `zero_grad` is one fill. Module parameters become views into
the arena (same names/shapes, so state_dict round-trips with the reference's checkpoints).
"""
import torch

from . import ops

ALIGN = 64  # floats (256 B): keeps every tensor 16-byte aligned for float4 access


def _pad4(n):
    return (n + 3) // 4 * 4


class ParamArena:
    def __init__(self, named_params, device, fused=None):
        """named_params: list of (name, nn.Parameter). Parameters are re-pointed at views of the arena.

        fused: optional {group_name: [param names]} of equally shaped 2-D weights (or 1-D biases) that are laid out
        back to back, each member padded to a multiple of 4 rows (elements), so that the whole group is ALSO one
        [sum(padded rows), cols] matrix (one vector): `self.fused[group_name] = (param_view, grad_view, padded_rows)`.
        One GEMM against that matrix then replaces one GEMM per member (the highway's N|L|G products, mu|logvar).
        The pad rows start at zero and stay zero: their gradients are written as zeros by the fused kernels, and Adam
        maps a zero gradient with zero moments to a zero update."""
        self.names = [n for n, _ in named_params]
        table = dict(named_params)
        member_of = {}
        for gname, members in (fused or {}).items():
            shapes = {tuple(table[m].shape) for m in members}
            if len(shapes) != 1 or len(next(iter(shapes))) not in (1, 2):
                raise ValueError("fused group %s needs equally shaped 1-D or 2-D members" % gname)
            for m in members:
                member_of[m] = gname
        self.offsets, off = {}, 0
        self._group_span = {}
        placed = set()

        def place(n, p, align):
            nonlocal off
            shape = tuple(p.shape)
            # big matrices whose rows are not a multiple of 4 floats (e.g. [*, 825] highway / LSTM input weights,
            # [*, 365] generator input weights) get a padded row stride so that GEMM operand rows are 16-byte aligned
            ld = shape[-1]
            if p.dim() == 2 and shape[0] >= 64 and shape[1] >= 64 and shape[1] % 4:
                ld = _pad4(shape[1])
            ne = p.numel() // shape[-1] * ld if p.dim() >= 1 and shape[-1] else p.numel()
            self.offsets[n] = (off, ne, shape, ld)
            if align:
                off += (ne + ALIGN - 1) // ALIGN * ALIGN
            else:  # fused member: pad to 4 rows (2-D) / 4 elements (1-D), no gap to the next member
                rows = _pad4(shape[0])
                off += rows * ld if p.dim() == 2 else rows
            return ld

        for n, p in named_params:
            if n in placed:
                continue
            g = member_of.get(n)
            if g is None:
                place(n, p, True)
                placed.add(n)
                continue
            start = off
            for m in fused[g]:
                ld = place(m, table[m], False)
                placed.add(m)
            shape = tuple(table[fused[g][0]].shape)
            self._group_span[g] = (start, off - start, _pad4(shape[0]), ld if len(shape) == 2 else None, shape)
            off = (off + ALIGN - 1) // ALIGN * ALIGN
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
        self.fused = {}
        for g, (start, ne, prow, ld, shape) in self._group_span.items():
            k = len(fused[g])

            def gview(flat):
                if ld is None:
                    return flat[start:start + ne]
                return flat[start:start + ne].view(k * prow, ld)[:, :shape[1]]

            self.fused[g] = (gview(self.p), gview(self.g), prow)

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
