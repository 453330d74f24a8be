"""Highway layer (reference: common/Highway.py:5-37):  x <- sigma(G x) * f(N x) + (1 - sigma(G x)) * (L x).

The three Linears of a layer run as ONE GEMM over the row-concatenated weight [3*out, in]; the gate
arithmetic is a fused epilogue kernel (K14).  f must be tanh (the only value used in the reference)."""
import torch
import torch.nn as nn

from .. import ops


class Highway(nn.Module):
    def __init__(self, input_size, output_size, num_layers=1, f=torch.tanh):
        super().__init__()
        if f is not torch.tanh:
            raise NotImplementedError("Highway on the HIP path implements f = tanh (the reference default)")
        self.num_layers = num_layers
        self.nonlinear = nn.ModuleList([nn.Linear(input_size, output_size) for _ in range(num_layers)])
        self.linear = nn.ModuleList([nn.Linear(input_size, output_size) for _ in range(num_layers)])
        self.gate = nn.ModuleList([nn.Linear(input_size, output_size) for _ in range(num_layers)])
        self.f = f
        self._packed = {}  # layer -> (key, w [3*out, in], b [3*out]): the row-concatenated weights, kept while the parameters stand still

    def _pack(self, i, g, n, l):
        """gate | nonlinear | linear rows as one matrix.  Under autograd the concatenation is part of the graph (its backward hands
        each Linear its rows); without gradients (inference, the only place a layer is called many times per weight update) the
        packed copy is kept until a parameter moves -- its ``_version`` / storage, or ops.PARAM_EPOCH for writes through ``.data``."""
        params = (g.weight, n.weight, l.weight, g.bias, n.bias, l.bias)
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return torch.cat(params[:3], dim=0), torch.cat(params[3:], dim=0)
        key = (ops.PARAM_EPOCH,) + tuple((p._version, p.data_ptr(), p.device) for p in params)
        hit = self._packed.get(i)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                hit = self._packed[i] = (key, torch.cat(params[:3], dim=0), torch.cat(params[3:], dim=0))
        return hit[1], hit[2]

    def forward(self, x):
        for i, (n, l, g) in enumerate(zip(self.nonlinear, self.linear, self.gate)):
            w, b = self._pack(i, g, n, l)  # [3*out, in]: gate | nonlinear | linear
            x = ops.highway_gate(ops.linear(x, w, b))
        return x
