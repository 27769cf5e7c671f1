"""Highway layers. Same class, constructor and parameter names as the reference's selfModules/highway.py:6-50
(`{nonlinear,linear,gate}_module_{i}-{0=weight,1=bias}`); forward/backward run on MFMA GEMMs with a fused gate kernel.
Only f = relu is implemented on the device (it is the only f the reference uses, model/encoder.py:15)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..hip import nn as hnn


def _linear_params(size):
    w = nn.Parameter(torch.empty(size, size))
    b = nn.Parameter(torch.empty(size))
    nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    bound = 1 / math.sqrt(size)
    nn.init.uniform_(b, -bound, bound)
    return [w, b]


class Highway(nn.Module):
    def __init__(self, size, num_layers, f):
        super(Highway, self).__init__()

        if f not in (F.relu, torch.relu):
            raise NotImplementedError("Highway on MI355X implements f = relu (the reference's only use)")
        self.num_layers = num_layers
        self.size = size

        # highway.py:13-23: three lists of Linear(size, size), parameters registered under dashed names
        self.nonlinear = [_linear_params(size) for _ in range(num_layers)]
        for i, module in enumerate(self.nonlinear):
            self._add_to_parameters(module, 'nonlinear_module_{}'.format(i))

        self.linear = [_linear_params(size) for _ in range(num_layers)]
        for i, module in enumerate(self.linear):
            self._add_to_parameters(module, 'linear_module_{}'.format(i))

        self.gate = [_linear_params(size) for _ in range(num_layers)]
        for i, module in enumerate(self.gate):
            self._add_to_parameters(module, 'gate_module_{}'.format(i))

        self.f = f

    def layer_params(self):
        """Per layer: dict(nw, nb, lw, lb, gw, gb) of Parameters (the layout hip/engine.py expects)."""
        return [dict(nw=self.nonlinear[l][0], nb=self.nonlinear[l][1], lw=self.linear[l][0], lb=self.linear[l][1],
                     gw=self.gate[l][0], gb=self.gate[l][1]) for l in range(self.num_layers)]

    def forward(self, x):
        """
        :param x: tensor with shape of [batch_size, size]
        :return: tensor with shape of [batch_size, size]
        σ(G x) ⨀ f(N x) + (1 - σ(G x)) ⨀ (L x) per layer (highway.py:38-44)
        """
        flat = []
        for p in self.layer_params():
            flat += [p[k] for k in ("nw", "nb", "lw", "lb", "gw", "gb")]
        return hnn._HighwayFn.apply(x, *flat)

    def _add_to_parameters(self, parameters, name):
        for i, parameter in enumerate(parameters):
            self.register_parameter(name='{}-{}'.format(name, i), param=parameter)
