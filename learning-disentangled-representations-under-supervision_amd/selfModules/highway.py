"""Highway layers on MI355X: per layer  y = g * relu(N x) + (1 - g) * (L x),  g = sigmoid(G x),  run as MFMA GEMMs with
a fused gate kernel (hip/nn.py `_HighwayFn`; the fused training step stacks N | L | G into one product, hip/engine.py).

Drop-in for the reference's selfModules/highway.py:6-50: class `Highway(size, num_layers, f)`, `forward(x[N, size])`,
and the reference's unusual parameter registry — the three affine maps of layer i are NOT sub-modules; their weight and
bias are registered directly on this module as `nonlinear_module_{i}-0/1`, `linear_module_{i}-0/1`,
`gate_module_{i}-0/1` (0 = weight, 1 = bias), all nonlinear ones first, then linear, then gate. Checkpoints depend on
those names and on that order, so both are kept. Only f = relu exists on the device (the reference's only use,
model/encoder.py:15).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..hip import nn as hnn

_BRANCHES = ("nonlinear", "linear", "gate")


def _affine(size):
    """Weight and bias initialised like nn.Linear(size, size) (what the reference instantiates)."""
    weight = nn.Parameter(torch.empty(size, size))
    bias = nn.Parameter(torch.empty(size))
    nn.init.kaiming_uniform_(weight, a=math.sqrt(5))
    nn.init.uniform_(bias, -1.0 / math.sqrt(size), 1.0 / math.sqrt(size))
    return weight, bias


class Highway(nn.Module):
    def __init__(self, size, num_layers, f):
        super().__init__()
        if f not in (F.relu, torch.relu):
            raise NotImplementedError("Highway on MI355X implements f = relu (the reference's only use)")
        self.size, self.num_layers, self.f = size, num_layers, f
        # branch-major creation and registration order (nonlinear 0..n-1, linear 0..n-1, gate 0..n-1)
        for branch in _BRANCHES:
            maps = [_affine(size) for _ in range(num_layers)]
            setattr(self, branch, maps)  # plain lists, like the reference's self.nonlinear / self.linear / self.gate
            for i, (weight, bias) in enumerate(maps):
                self.register_parameter('%s_module_%d-0' % (branch, i), weight)
                self.register_parameter('%s_module_%d-1' % (branch, i), bias)

    def layer_params(self):
        """Per layer: dict(nw, nb, lw, lb, gw, gb) of Parameters (the layout hip/engine.py expects)."""
        out = []
        for i in range(self.num_layers):
            (nw, nb), (lw, lb), (gw, gb) = self.nonlinear[i], self.linear[i], self.gate[i]
            out.append(dict(nw=nw, nb=nb, lw=lw, lb=lb, gw=gw, gb=gb))
        return out

    def forward(self, x):
        """x [N, size] -> [N, size]."""
        flat = [p[k] for p in self.layer_params() for k in ("nw", "nb", "lw", "lb", "gw", "gb")]
        return hnn._HighwayFn.apply(x, *flat)
