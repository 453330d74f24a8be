"""TransformerBlock on the HIP path (reference: common/TransformerBlock.py:7-32).

    r = x + drop(MHA(LN1(x)))      residual from the UN-normed input (:27)
    y = W2 drop(relu(W1 LN2(r)))   no residual, no activation after W2 (:28-29)
    y[pad] = 0                     (:31)
This block at width 5H (head_dim 5H/8) carries ~73 % of CaSE's forward FLOPs.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import config, ops
from .attention import MultiheadAttention


class TransformerBlock(nn.Module):
    def __init__(self, num_heads, input_hidden_size, output_hidden_size, activation=None):
        super().__init__()
        self.output_hidden_size = output_hidden_size
        self.self_attn = MultiheadAttention(input_hidden_size, num_heads, dropout=0.1)
        self.norm1 = nn.LayerNorm(input_hidden_size)
        self.norm2 = nn.LayerNorm(input_hidden_size)
        self.linear1 = nn.Linear(input_hidden_size, output_hidden_size)
        self.linear2 = nn.Linear(output_hidden_size, output_hidden_size)
        if activation is None or activation is F.relu:
            self.activation = "relu"
        elif activation is F.gelu:
            self.activation = "gelu"
        else:
            raise NotImplementedError("TransformerBlock on the HIP path supports relu (reference default) or gelu")

    def forward(self, input, input_mask):
        """input [B, N, L, Ein]; input_mask [B, N, L] bool, True = token.  Returns [B, N, L, Eout]."""
        B, N, L, E = input.shape
        x = input.reshape(B * N, L, E)
        valid = input_mask.reshape(B * N, L)
        p = config.drop_p(0.1, self.training)
        if torch.is_grad_enabled() and x.requires_grad:
            # x + MHA(LN(x)): the residual gradient of x is added inside the LayerNorm backward kernel; when x is the Interaction's
            # concatenation itself, that kernel also runs the concatenation's backward (the 5H-wide gradient is never written)
            fused = ops.concat5_layer_norm_carry(input, self.norm1.weight, self.norm1.bias, self.norm1.eps)
            if fused is not None:
                n1, xres = fused[0].reshape(B * N, L, E), fused[1].reshape(B * N, L, E)
            else:
                n1, xres = ops.layer_norm_carry(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        else:
            n1, xres = ops.layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps), x
        # out-projection + dropout + residual -> LN2 as ONE op: its backward emits the dropout-masked gradient from the LayerNorm kernel
        n2 = self.self_attn.self_attention(n1, valid, residual=xres, p_res=0.1, ln=(self.norm2.weight, self.norm2.bias, self.norm2.eps))
        y = ops.ffn(n2, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias,
                    self.activation, p_inner=p, p_out=0.0)
        y = ops.mask_rows(y, valid, in_place=True)  # (y is this block's own FFN output)
        return y.reshape(B, N, L, self.output_hidden_size)
