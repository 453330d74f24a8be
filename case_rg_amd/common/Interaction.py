"""Query/passage dual co-attention (reference: common/Interaction.py:5-75).

With w = [w1; w2; w3] = dual_att_linear.weight:
    U[n,i,j] = w1.Eq[j] + w2.Ep[i] + (w3 * Ep[i]).Eq[j]
which equals the reference's Linear(cat[Eq, Ep, Eq*Ep]) (:32-36) without the [n, Lp, Lq, 3H] tensor (48 GB at
cfg 2).  U and its transpose come from two small MFMA GEMMs (rank-1 terms in the epilogue), the row softmaxes
of both give A = softmax_j and Bm^T = softmax_i directly in the layouts the four follow-up bmm need:
    A1 = A Eq, B1 = Bm^T Ep, A2 = A B1, B2 = Bm^T A1
    G_q_p = [Ep, A1, A2, Ep*A1, Ep*A2] (0 at pads), G_p_q = [Eq, B1, B2, Eq*B1, Eq*B2] (0 at pads; max over
    passages when one query faces P passages, :73-74).
"""
import torch
import torch.nn as nn

from .. import ops


class Interaction(nn.Module):
    def __init__(self, hidden_size):
        super().__init__()
        self.hidden_size = hidden_size
        self.dual_att_linear = nn.Linear(3 * hidden_size, 1, bias=False)

    def forward(self, encode_input1, encode_input2, input1_mask, input2_mask):
        B, nq, Lq, H = encode_input1.shape
        _, P, Lp, _ = encode_input2.shape
        needs_grad = torch.is_grad_enabled() and (encode_input1.requires_grad or encode_input2.requires_grad or self.dual_att_linear.weight.requires_grad)
        if ops.interaction_supported(encode_input1, encode_input2, needs_grad):  # K8 as two kernels (csrc/interaction.hip)
            G_p_q, G_q_p, _, _ = ops.interaction_fwd(encode_input1, encode_input2, input1_mask, input2_mask, self.dual_att_linear.weight)
            return (ops.max_over_p(G_p_q) if nq != P else G_p_q), G_q_p
        if needs_grad and ops.interaction_train_supported(encode_input1, encode_input2):  # the same kernels with an explicit backward
            Ep_g, Ep_i = ops.fanout(encode_input2, 2)
            G_p_q, G_q_p, A1p, A2p = ops.InteractionFn.apply(encode_input1, Ep_i, self.dual_att_linear.weight, input1_mask, input2_mask)
            # the first TransformerBlock's LayerNorm<5H> takes its gradient straight to the pieces (ops.concat5_layer_norm_carry)
            # (on an ALIAS of the output: the pieces are outputs of the same autograd node, which saves G_q_p for its backward -- hung on
            #  G_q_p itself they would close a reference cycle through that node and every step's graph would stay alive)
            G = G_q_p.view_as(G_q_p)
            G._case_concat5 = (Ep_g, A1p, A2p, ops._u8(input2_mask.reshape(-1, Lp).contiguous()))
            return (ops.max_over_p(G_p_q) if nq != P else G_p_q), G
        if nq != P:
            assert nq == 1
            Eq = encode_input1.expand(-1, P, -1, -1)
            qv = input1_mask.expand(-1, P, -1)
        else:
            Eq, qv = encode_input1, input1_mask
        n = B * P
        Eq = Eq.reshape(n, Lq, H)
        Ep = encode_input2.reshape(n, Lp, H)
        qv = qv.reshape(n, Lq).contiguous()
        pv = input2_mask.reshape(n, Lp).contiguous()
        w = self.dual_att_linear.weight
        w1, w2, w3 = w[:, :H], w[:, H:2 * H], w[0, 2 * H:]
        f32 = torch.float32
        # every tensor below feeds 2 - 5 products: ops.fanout hands each consumer its own alias, so that the backward pass sums the
        # gradients of one tensor in ONE kernel (autograd would add them pairwise: 12 elementwise launches per call on [n, L, H] tensors)
        Ep_w2, Ep_w3, Ep_b1, Ep_g = ops.fanout(Ep, 4)
        Eq_w1, Eq_u, Eq_ut, Eq_a1, Eq_g = ops.fanout(Eq, 5)
        cq = ops.linear(Eq_w1, w1, out_dtype=f32).reshape(n, Lq)  # w1 . Eq[j]
        ap = ops.linear(Ep_w2, w2, out_dtype=f32).reshape(n, Lp)  # w2 . Ep[i]
        Epw_u, Epw_ut = ops.fanout(ops.scale_cols(Ep_w3, w3), 2)
        U = ops.bmm(Epw_u, Eq_u, bias_row=ap, out_dtype=f32) + cq.unsqueeze(1)     # [n, Lp, Lq]
        Ut = ops.bmm(Eq_ut, Epw_ut, bias_row=cq, out_dtype=f32) + ap.unsqueeze(1)  # [n, Lq, Lp]
        dt = Ep.dtype
        A_1, A_2 = ops.fanout(ops.masked_softmax(U, qv, pv, outer=n, out_dtype=dt), 2)     # softmax over the query axis
        Bt_1, Bt_2 = ops.fanout(ops.masked_softmax(Ut, pv, qv, outer=n, out_dtype=dt), 2)  # softmax over the passage axis, transposed
        A1_b, A1_g = ops.fanout(ops.bmm(A_1, Eq_a1, b_is_kn=True), 2)    # [n, Lp, H]
        B1_a, B1_g = ops.fanout(ops.bmm(Bt_1, Ep_b1, b_is_kn=True), 2)   # [n, Lq, H]
        A2 = ops.bmm(A_2, B1_a, b_is_kn=True)
        B2 = ops.bmm(Bt_2, A1_b, b_is_kn=True)
        G_q_p = ops.concat5(Ep_g, A1_g, A2, pv, shape=(B, P, Lp, 5 * H))
        G_p_q = ops.concat5(Eq_g, B1_g, B2, qv, shape=(B, P, Lq, 5 * H))
        if nq != P:
            G_p_q = ops.max_over_p(G_p_q)
        return G_p_q, G_q_p
