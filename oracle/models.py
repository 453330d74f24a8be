"""Oracle L2/L3: sequence encoder/decoder stacks and the CaSE / Masque task models.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  fp32 CPU restatement with the reference's
module-attribute graph (so ``state_dict`` keys match) and dropout-free maths.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .blocks import (BilinearAttention, Interaction, PositionalEmbedding, TransformerBlock,
                     TransformerDecoder, TransformerDecoderLayer, TransformerEncoder,
                     TransformerEncoderLayer, causal_additive_mask, masked_mean, one_hot_map)

# reference: common/Constants.py:1-7
SPECIALS = dict(PAD="[PAD]", BOS="[unused0]", UNK="[UNK]", EOS="[unused1]", SEP="[SEP]",
                CLS="[CLS]", MASK="[MASK]")


def ids_to_tokens(batch_ids, id2vocab):
    """Drop BOS/PAD, stop at EOS, empty -> [UNK].  Reference: common/Utils.py:200-217."""
    out = []
    for row in batch_ids:
        words = []
        for i in row:
            w = id2vocab[int(i)]
            if w in (SPECIALS["BOS"], SPECIALS["PAD"]):
                continue
            if w == SPECIALS["EOS"]:
                break
            words.append(w)
        out.append(words or [SPECIALS["UNK"]])
    return out


def _embedding(vocab, width, max_len=1000):
    return nn.Sequential(nn.Embedding(vocab, width, padding_idx=0),
                         PositionalEmbedding(width, dropout=0.1, max_len=max_len))


# --------------------------------------------------------------------------------------------
# common/TransformerSeqEncoderDecoder.py:14-45
# --------------------------------------------------------------------------------------------
class TransformerSeqEncoder(nn.Module):
    """ids [B, N, L] -> (out [B, N, 1, L, H], state [B, N, 1, H]); embed*sqrt(H)+pos -> N-layer
    encoder (gelu, FFN width = H) with key padding -> masked mean."""

    def __init__(self, num_layers, num_heads, src_vocab_size, hidden_size, emb_matrix=None, norm=None):
        super().__init__()
        self.num_layers, self.num_heads = num_layers, num_heads
        self.embedding = _embedding(src_vocab_size, hidden_size)
        self.enc = TransformerEncoder(
            TransformerEncoderLayer(hidden_size, nhead=num_heads, dim_feedforward=hidden_size,
                                    dropout=0.1, activation="gelu"),
            num_layers=num_layers, norm=norm)

    def forward(self, batch_numseq_seqlen):
        B, N, L = batch_numseq_seqlen.shape
        ids = batch_numseq_seqlen.reshape(B * N, L)
        valid = ids.ne(0)
        x = self.embedding(ids)
        y = self.enc(x.transpose(0, 1), src_key_padding_mask=~valid).transpose(0, 1)
        state = masked_mean(y, valid)
        return y.reshape(B, N, L, -1).unsqueeze(2), state.reshape(B, N, -1).unsqueeze(2)


# --------------------------------------------------------------------------------------------
# Pointer-generator decoders.  One shared implementation parameterised by the three variants:
#   generic  common/TransformerSeqEncoderDecoder.py:47-150
#   CaSE     CaSE/Model.py:13-125
#   Masque   Masque/Model.py:13-119
# --------------------------------------------------------------------------------------------
class _PointerDecoderBase(nn.Module):
    def _stacks(self, num_memories, num_layers, nhead, hidden_size):
        self.decs = nn.ModuleList([
            TransformerDecoder(TransformerDecoderLayer(hidden_size, nhead=nhead, dim_feedforward=hidden_size,
                                                       dropout=0.1, activation="gelu"),
                               num_layers=num_layers, norm=None)
            for _ in range(num_memories)])

    # -- one pass over a (teacher-forced or greedy-prefix) decoder input ----------------------
    def _run_prefix(self, dec_ids, memories, mem_valid, mem_weights, feature):
        """Chain decs[0] -> attns[0] -> decs[1] -> attns[1] ...  Returns (dec_in, dec_out_pre_norm,
        contexts, copy attentions).  CaSE/Model.py:66-83."""
        T = dec_ids.size(1)
        dec_in = self.embedding(dec_ids)
        tgt_valid = dec_ids.ne(0)
        x = dec_in.transpose(0, 1)
        ctxs, copies = [], []
        for i, mem in enumerate(memories):
            x, _, _ = self.decs[i](x, mem.transpose(0, 1), tgt_mask=causal_additive_mask(T),
                                   tgt_key_padding_mask=~tgt_valid,
                                   memory_key_padding_mask=~mem_valid[i])
            q = x.transpose(0, 1)
            if feature is not None:
                q = torch.cat([q, feature], dim=-1)
            pair = tgt_valid[:, :, None] & mem_valid[i][:, None, :]
            ctx, _, p = self.attns[i](q, mem, mem, mask=pair)
            ctxs.append(ctx)
            if mem_weights is not None:
                p = mem_weights[i][:, None, :] * p
                p = p / (1e-8 + p.sum(dim=-1, keepdim=True))
            copies.append(p)
        return dec_in, x, ctxs, copies

    def _mix(self, dec_out, ctxs, gen, copies, source_map):
        """p = softmax(mix([dec_out, ctx...])); dist1 = p0*gen; dist2 = cat_k(p_k*copy_k) @ onehot.
        CaSE/Model.py:38-48."""
        p = torch.softmax(self.mix(torch.cat([dec_out] + ctxs, dim=-1)), dim=-1)
        dist1 = p[:, :, 0:1] * gen
        ptr = torch.cat([p[:, :, k + 1:k + 2] * c for k, c in enumerate(copies)], dim=-1)
        return dist1, ptr @ source_map


class CaSETransformerSeqDecoder(_PointerDecoderBase):
    def __init__(self, num_memories, num_layers, nhead, tgt_vocab_size, hidden_size, emb_matrix=None):
        super().__init__()
        H = hidden_size
        self.tgt_vocab_size, self.num_layers, self.hidden_size = tgt_vocab_size, num_layers, H
        self.embedding = _embedding(tgt_vocab_size, H)
        self._stacks(num_memories, num_layers, nhead, H)
        self.norm1 = nn.LayerNorm(H)
        self.norm2 = nn.LayerNorm(H)
        self.attns = nn.ModuleList([BilinearAttention(2 * H, H, H) for _ in range(num_memories)])
        self.gen = nn.Sequential(nn.Linear(3 * H, H), nn.Identity(), nn.Linear(H, tgt_vocab_size, bias=False),
                                 nn.Softmax(dim=-1))
        self.mix = nn.Linear(3 * H, num_memories + 1)

    def _step(self, dec_ids, memories, mem_valid, mem_weights, answer_feature, source_map):
        feat = self.norm2(answer_feature)[:, None, :].expand(-1, dec_ids.size(1), -1)
        dec_in, x, ctxs, copies = self._run_prefix(dec_ids, memories, mem_valid, mem_weights, feat)
        dec_out = self.norm1(x).transpose(0, 1)
        gen = self.gen(torch.cat([dec_in, dec_out, feat], dim=-1))
        dist1, dist2 = self._mix(dec_out, ctxs, gen, copies, source_map)
        return dec_out, gen, dist1, dist2

    def forward(self, encode_memories, BOS, UNK, source_map, groundtruth_index=None,
                additional_decoder_feature=None, encode_weights=None, encode_masks=None,
                init_decoder_state=None, max_target_length=None):
        B = source_map.size(0)
        H = self.hidden_size
        weights = [w.reshape(B, -1) for w in encode_weights]
        mems = [m.reshape(B, -1, H) for m in encode_memories]
        valid = [m.reshape(B, -1) for m in encode_masks]
        if max_target_length is None:
            max_target_length = groundtruth_index.size(1)
        bos = torch.full((B, 1), BOS, dtype=torch.long)
        if self.training and groundtruth_index is not None:
            dec_ids = torch.cat([bos, groundtruth_index[:, :-1]], dim=-1)
            dec_out, gen, d1, d2 = self._step(dec_ids, mems, valid, weights, additional_decoder_feature, source_map)
            return dec_out, gen, (d1, d2), groundtruth_index
        # greedy: the whole prefix is re-run every step (CaSE/Model.py:94-123); argmax of the last
        # position with lowest-index tie break (common/Utils.py:167)
        picked = []
        for _ in range(max_target_length):
            dec_ids = torch.cat([bos] + picked, dim=-1)
            dec_out, gen, d1, d2 = self._step(dec_ids, mems, valid, weights, additional_decoder_feature, source_map)
            dist = d1 + d2
            picked.append(dist[:, -1].max(dim=1, keepdim=True)[1])
        return dec_out, gen, dist, torch.cat(picked, dim=-1)


class MasqueTransformerSeqDecoder(_PointerDecoderBase):
    def __init__(self, num_memories, num_layers, nhead, tgt_vocab_size, hidden_size, emb_matrix=None):
        super().__init__()
        H = hidden_size
        self.tgt_vocab_size, self.num_layers, self.hidden_size = tgt_vocab_size, num_layers, H
        self.embedding = _embedding(tgt_vocab_size, H)
        self._stacks(num_memories, num_layers, nhead, H)
        self.norm = nn.LayerNorm(H)
        self.attns = nn.ModuleList([BilinearAttention(H, H, H) for _ in range(num_memories)])
        self.gen = nn.Sequential(nn.Linear(2 * H, H), nn.Linear(H, tgt_vocab_size, bias=False), nn.Softmax(dim=-1))
        self.mix = nn.Linear(3 * H, num_memories + 1)

    def _step(self, dec_ids, memories, mem_valid, mem_weights, source_map):
        dec_in, x, ctxs, copies = self._run_prefix(dec_ids, memories, mem_valid, mem_weights, None)
        dec_out = self.norm(x).transpose(0, 1)
        gen = self.gen(torch.cat([dec_in, dec_out], dim=-1))
        d1, d2 = self._mix(dec_out, ctxs, gen, copies, source_map)
        return dec_out, gen, d1 + d2

    def forward(self, encode_memories, BOS, UNK, source_map, encode_masks=None, encode_weights=None,
                groundtruth_index=None, init_decoder_state=None, max_target_length=None):
        B = source_map.size(0)
        H = self.hidden_size
        weights = None if encode_weights is None else [w.reshape(B, -1) for w in encode_weights]
        mems = [m.reshape(B, -1, H) for m in encode_memories]
        valid = [m.reshape(B, -1) for m in encode_masks]
        if max_target_length is None:
            max_target_length = groundtruth_index.size(1)
        bos = torch.full((B, 1), BOS, dtype=torch.long)
        if self.training and groundtruth_index is not None:
            dec_ids = torch.cat([bos, groundtruth_index[:, :-1]], dim=-1)
            dec_out, gen, dist = self._step(dec_ids, mems, valid, weights, source_map)
            return dec_out, gen, dist, groundtruth_index
        picked = []
        for _ in range(max_target_length):
            dec_ids = torch.cat([bos] + picked, dim=-1)
            dec_out, gen, dist = self._step(dec_ids, mems, valid, weights, source_map)
            picked.append(dist[:, -1].max(dim=1, keepdim=True)[1])
        return dec_out, gen, dist, torch.cat(picked, dim=-1)


class TransformerSeqDecoder(MasqueTransformerSeqDecoder):
    """Generic variant (common/TransformerSeqEncoderDecoder.py:47-150): same maths as Masque's but takes
    a *list* of source maps (concatenated along the source axis, :66) and mix width H + M*H."""

    def __init__(self, num_memories, num_layers, nhead, tgt_vocab_size, hidden_size, emb_matrix=None):
        super().__init__(num_memories, num_layers, nhead, tgt_vocab_size, hidden_size)
        self.mix = nn.Linear(hidden_size + num_memories * hidden_size, num_memories + 1)

    def forward(self, encode_memories, BOS, UNK, source_maps, encode_masks=None, encode_weights=None,
                groundtruth_index=None, init_decoder_state=None, max_target_length=None):
        return super().forward(encode_memories, BOS, UNK, torch.cat(source_maps, dim=-2),
                               encode_masks=encode_masks, encode_weights=encode_weights,
                               groundtruth_index=groundtruth_index, max_target_length=max_target_length)


# --------------------------------------------------------------------------------------------
# Selection / span / generation heads
# --------------------------------------------------------------------------------------------
def _blocks(heads, H, n_after_first):
    return nn.ModuleList([TransformerBlock(heads, 5 * H, H)] +
                         [TransformerBlock(heads, H, H) for _ in range(n_after_first)])


def _run_blocks(blocks, x, mask):
    for b in blocks:
        x = b(x, mask)
    return x


class RelevantPassageSelection(nn.Module):
    """CaSE/Model.py:127-163.  Interaction -> 3 query / 5 passage blocks -> Linear(H,1) on [CLS]."""

    def __init__(self, hidden_size, num_heads, query_encoder, passage_encoder):
        super().__init__()
        self.hidden_size, self.num_heads = hidden_size, num_heads
        self.query_encoder, self.passage_encoder = query_encoder, passage_encoder
        self.interaction = Interaction(hidden_size)
        self.query_blocks = _blocks(num_heads, hidden_size, 2)
        self.passage_blocks = _blocks(num_heads, hidden_size, 4)
        self.scorer = nn.Linear(hidden_size, 1)

    def action(self, query, passage, encode_query, encode_passage):
        eq, ep = encode_query[0][:, :, -1], encode_passage[0][:, :, -1]
        pm, qm = passage.ne(0), query.ne(0)
        g_pq, g_qp = self.interaction(eq, ep, qm, pm)
        qr = _run_blocks(self.query_blocks, g_pq, qm)
        pr = _run_blocks(self.passage_blocks, g_qp, pm)
        score = self.scorer(pr[:, :, 0]).squeeze(-1)
        return score, (qr, qr[:, :, 0]), (pr, pr[:, :, 0])


class PassageSelection(RelevantPassageSelection):
    """Masque/Model.py:121-159: same network, returns bare tensors and encodes on demand."""

    def action(self, query, passage, encode_query=None, encode_passage=None):
        if encode_query is None:
            encode_query = self.query_encoder(query)[0][:, :, -1]
        if encode_passage is None:
            encode_passage = self.passage_encoder(passage)[0][:, :, -1]
        pm, qm = passage.ne(0), query.ne(0)
        g_pq, g_qp = self.interaction(encode_query, encode_passage, qm, pm)
        qr = _run_blocks(self.query_blocks, g_pq, qm)
        pr = _run_blocks(self.passage_blocks, g_qp, pm)
        return self.scorer(pr[:, :, 0]).squeeze(-1), qr, pr


class SupportingTokenIdentification(nn.Module):
    """CaSE/Model.py:165-212."""

    def __init__(self, max_span_size, hidden_size, num_heads, query_encoder, passage_encoder, passage_selection):
        super().__init__()
        self.hidden_size, self.num_heads, self.max_span_size = hidden_size, num_heads, max_span_size
        self.query_encoder, self.passage_encoder = query_encoder, passage_encoder
        self.passage_selection = passage_selection
        self.interaction = Interaction(hidden_size)
        self.query_blocks = _blocks(num_heads, hidden_size, 1)
        self.passage_blocks = _blocks(num_heads, hidden_size, 2)
        self.norm1 = nn.LayerNorm(hidden_size)
        self.norm2 = nn.LayerNorm(hidden_size)
        self.scorer = nn.Linear(hidden_size, 1)

    def action(self, query, passage, encode_query, encode_passage, passage_selection_result):
        pm, qm = passage.ne(0), query.ne(0)
        _, q1, p1 = passage_selection_result
        g_pq, g_qp = self.interaction(q1[0], p1[0], qm, pm)
        qr = _run_blocks(self.query_blocks, g_pq, qm)
        pr = _run_blocks(self.passage_blocks, g_qp, pm)
        tok = self.scorer(pr).squeeze(-1).masked_fill(~pm, -1e6).clamp(min=-1e6, max=1e6)
        qr = self.norm1(q1[0] + qr)
        pr = self.norm2(p1[0] + pr)
        return tok, (qr, qr[:, :, 0]), (pr, pr[:, :, 0])


class CaSEResponseGeneration(nn.Module):
    """CaSE/Model.py:214-253."""

    def __init__(self, BOS, UNK, vocab_size, hidden_size, num_heads, query_encoder, passage_encoder,
                 passage_selection, span_extraction, decoder):
        super().__init__()
        self.hidden_size, self.vocab_size, self.num_heads = hidden_size, vocab_size, num_heads
        self.query_encoder, self.passage_encoder = query_encoder, passage_encoder
        self.passage_selection, self.span_extraction = passage_selection, span_extraction
        self.BOS, self.UNK = BOS, UNK
        self.decoder = decoder

    def action(self, query, passage, source_map, encode_query, encode_passage, passage_selection_result,
               span_extraction_result, output=None, max_target_length=None):
        B = query.size(0)
        p_score = passage_selection_result[0]
        t_score, q_rep, p_rep = span_extraction_result
        prior = (torch.sigmoid(p_score)[..., None] * torch.sigmoid(t_score)).reshape(B, -1)
        prior = prior / (1e-8 + prior.sum(dim=-1, keepdim=True))
        answer_rep = (prior[:, None, :] @ p_rep[0].reshape(B, -1, p_rep[0].size(-1))).squeeze(1)
        prior_p = prior.reshape_as(t_score)
        prior_q = torch.ones(B, 1, q_rep[0].size(2))
        return self.decoder([q_rep[0], p_rep[0]], self.BOS, self.UNK, source_map,
                            additional_decoder_feature=answer_rep, groundtruth_index=output,
                            max_target_length=max_target_length,
                            encode_masks=[query.ne(0), passage.ne(0)], encode_weights=[prior_q, prior_p])


class MasqueResponseGeneration(nn.Module):
    """Masque/Model.py:161-200."""

    def __init__(self, BOS, UNK, vocab_size, hidden_size, num_heads, query_encoder, passage_encoder,
                 passage_selection, decoder):
        super().__init__()
        self.hidden_size, self.vocab_size, self.num_heads = hidden_size, vocab_size, num_heads
        self.query_encoder, self.passage_encoder = query_encoder, passage_encoder
        self.passage_selection = passage_selection
        self.BOS, self.UNK = BOS, UNK
        self.decoder = decoder

    def action(self, query, passage, source_map, encode_query=None, encode_passage=None,
               passage_selection_result=None, output=None, max_target_length=None):
        if encode_query is None:
            encode_query = self.query_encoder(query)[0][:, :, -1]
        if encode_passage is None:
            encode_passage = self.passage_encoder(passage)[0][:, :, -1]
        if passage_selection_result is None:
            passage_selection_result = self.passage_selection.action(query, passage, encode_query, encode_passage)
        p_score, q_rep, p_rep = passage_selection_result
        B = query.size(0)
        prior_q = torch.ones(B, 1, q_rep.size(2))
        prior_p = torch.sigmoid(p_score)[..., None].expand(-1, -1, p_rep.size(2))
        return self.decoder([q_rep, p_rep], self.BOS, self.UNK, source_map, groundtruth_index=output,
                            max_target_length=max_target_length,
                            encode_masks=[query.ne(0), passage.ne(0)], encode_weights=[prior_q, prior_p])


def _passage_bce(score, label_index):
    target = torch.zeros_like(score).scatter_(1, label_index[:, None], 1.0)
    return F.binary_cross_entropy_with_logits(score, target).unsqueeze(0)


def _nll(dist, target):
    V = dist.size(-1)
    return F.nll_loss((dist + 1e-8).log().reshape(-1, V), target.reshape(-1), ignore_index=0).unsqueeze(0)


class CaSE(nn.Module):
    """CaSE/Model.py:255-339.  enc_layers / dec_layers / heads are exposed (reference hard-codes 3 / 4 / 8)."""

    def __init__(self, max_span_size, max_target_length, id2vocab, vocab2id, hidden_size,
                 enc_layers=3, dec_layers=4, heads=8):
        super().__init__()
        V = len(vocab2id)
        self.UNK = vocab2id[SPECIALS["UNK"]]
        self.max_target_length = max_target_length
        self.query_encoder = TransformerSeqEncoder(enc_layers, heads, V, hidden_size)
        self.passage_encoder = self.query_encoder
        self.passage_selection = RelevantPassageSelection(hidden_size, heads, self.query_encoder, self.passage_encoder)
        self.span_extraction = SupportingTokenIdentification(max_span_size, hidden_size, heads, self.query_encoder,
                                                             self.passage_encoder, self.passage_selection)
        self.response_generation = CaSEResponseGeneration(
            vocab2id[SPECIALS["BOS"]], self.UNK, V, hidden_size, heads, self.query_encoder, self.passage_encoder,
            self.passage_selection, self.span_extraction,
            CaSETransformerSeqDecoder(2, dec_layers, heads, V, hidden_size))
        self.id2vocab, self.vocab2id, self.vocab_size = id2vocab, vocab2id, len(id2vocab)

    def to_sentence(self, data, batch_indices):
        return ids_to_tokens(batch_indices, self.id2vocab)

    def _encode_and_select(self, data):
        eq, ep = self.query_encoder(data["query"]), self.passage_encoder(data["passage"])
        ps = self.passage_selection.action(data["query"], data["passage"], encode_query=eq, encode_passage=ep)
        se = self.span_extraction.action(data["query"], data["passage"], encode_query=eq, encode_passage=ep,
                                         passage_selection_result=ps)
        return eq, ep, ps, se

    def do_train(self, data):
        eq, ep, ps, se = self._encode_and_select(data)
        loss_ps = _passage_bce(ps[0], data["passage_label"])
        valid = data["passage"].ne(0).float()
        bce = F.binary_cross_entropy_with_logits(se[0], data["token_label"], reduction="none")
        loss_se = (valid * bce * data["token_weight"]).sum() / valid.sum()
        rg = self.response_generation.action(data["query"], data["passage"], data["source_map"], eq, ep, ps, se,
                                             output=data["response"])
        d1, d2 = rg[2]
        return [loss_ps, loss_se, _nll(d1 + d2, data["response"])]

    def do_test(self, data):
        eq, ep, ps, se = self._encode_and_select(data)
        rg = self.response_generation.action(data["query"], data["passage"], data["source_map"], eq, ep, ps, se,
                                             output=None, max_target_length=self.max_target_length)
        return {"answer": rg[3], "rank": ps[0]}

    def forward(self, data, method="mle_train"):
        if "source_map" in data:
            data["source_map"] = one_hot_map(data["source_map"], self.vocab_size)
        if method == "train":
            return self.do_train(data)
        if method == "test":
            return self.do_test(data)


class Masque(nn.Module):
    """Masque/Model.py:202-286."""

    def __init__(self, max_target_length, id2vocab, vocab2id, hidden_size, enc_layers=3, dec_layers=4, heads=8):
        super().__init__()
        V = len(vocab2id)
        self.UNK = vocab2id[SPECIALS["UNK"]]
        self.max_target_length = max_target_length
        self.query_encoder = TransformerSeqEncoder(enc_layers, heads, V, hidden_size)
        self.passage_encoder = self.query_encoder
        self.passage_selection = PassageSelection(hidden_size, heads, self.query_encoder, self.passage_encoder)
        self.response_generation = MasqueResponseGeneration(
            vocab2id[SPECIALS["BOS"]], self.UNK, V, hidden_size, heads, self.query_encoder, self.passage_encoder,
            self.passage_selection, MasqueTransformerSeqDecoder(2, dec_layers, heads, V, hidden_size))
        self.id2vocab, self.vocab2id, self.vocab_size = id2vocab, vocab2id, len(id2vocab)

    def to_sentence(self, data, batch_indices):
        return ids_to_tokens(batch_indices, self.id2vocab)

    def _encode_and_select(self, data):
        eq = self.query_encoder(data["query"])[0][:, :, -1]
        ep = self.passage_encoder(data["passage"])[0][:, :, -1]
        return eq, ep, self.passage_selection.action(data["query"], data["passage"], eq, ep)

    def do_train(self, data):
        eq, ep, ps = self._encode_and_select(data)
        rg = self.response_generation.action(data["query"], data["passage"], data["source_map"], eq, ep, ps,
                                             output=data["response"])
        return [0.25 * _passage_bce(ps[0], data["passage_label"]), _nll(rg[2], data["response"])]

    def do_ps_train(self, data):
        _, _, ps = self._encode_and_select(data)
        return [_passage_bce(ps[0], data["passage_label"])]

    def do_test(self, data):
        eq, ep, ps = self._encode_and_select(data)
        rg = self.response_generation.action(data["query"], data["passage"], data["source_map"], eq, ep, ps,
                                             output=None, max_target_length=self.max_target_length)
        return {"answer": rg[3], "rank": ps[0]}

    def forward(self, data, method="mle_train"):
        data["source_map"] = one_hot_map(data["source_map"], self.vocab_size)
        if method == "train":
            return self.do_train(data)
        if method == "ps_train":
            return self.do_ps_train(data)
        if method == "test":
            return self.do_test(data)
