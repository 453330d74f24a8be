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

    def forward(self, x):
        for n, l, g in zip(self.nonlinear, self.linear, self.gate):
            w = torch.cat([g.weight, n.weight, l.weight], dim=0)  # [3*out, in]: gate | nonlinear | linear
            b = torch.cat([g.bias, n.bias, l.bias], dim=0)
            x = ops.highway_gate(ops.linear(x, w, b))
        return x
