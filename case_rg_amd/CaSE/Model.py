"""CaSE task model on the HIP path (reference: CaSE/Model.py:13-339).

Module-attribute graph, constructor signatures and ``state_dict`` keys follow the reference (the shared encoder is
reachable under 16 prefixes, 1301 keys), so reference checkpoints load with ``strict=True``
and the reference's ``CumulativeTrainer`` / ``Run.py`` drive this class unchanged.  ``enc_layers`` /
``dec_layers`` / ``heads`` expose the counts the reference hard-codes (3 / 4 / 8).
"""
import torch
import torch.nn as nn

from .. import config, ops
from ..common.Constants import BOS_WORD, EOS_WORD, UNK_WORD
from ..common.Interaction import Interaction
from ..common.TransformerSeqEncoderDecoder import PointerDecoderCore, TransformerSeqEncoder
from ..common.Utils import to_sentence
from ..common.heads import block_stack, generation_nll, passage_bce, run_block_pair, run_blocks


class CaSETransformerSeqDecoder(PointerDecoderCore):
    """Two-memory pointer-generator decoder conditioned on the answer representation (reference :13-125)."""

    def __init__(self, num_memories, num_layers, nhead, tgt_vocab_size, hidden_size, emb_matrix=None):
        super().__init__()
        H = hidden_size
        self._build(num_memories, num_layers, nhead, tgt_vocab_size, H, 2 * H, emb_matrix=emb_matrix)
        self.norm1 = nn.LayerNorm(H)
        self.norm2 = nn.LayerNorm(H)
        self.gen = nn.Sequential(nn.Linear(3 * H, H), nn.Dropout(0.1), nn.Linear(H, tgt_vocab_size, bias=False), nn.Softmax(dim=-1))
        self.mix = nn.Linear(3 * H, num_memories + 1)

    def extend(self, dec_outputs, gen_outputs, memory_weights, source_map):
        """Public form of the mixing step (reference :38-48); ``dec_outputs`` is cat[dec_out, ctx_q, ctx_p]."""
        H = self.hidden_size
        d1, d2 = self._mix(dec_outputs[..., :H], [dec_outputs[..., H:2 * H], dec_outputs[..., 2 * H:]], gen_outputs,
                           memory_weights, source_map)
        return (d1, d2) if self.training else d1 + d2

    def _feature(self, answer_rep, T):
        """LN2(answer_rep) broadcast over the T decoder positions, dropout 0.1 in training (reference :69, :98)."""
        feat = ops.layer_norm(answer_rep, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        return ops.dropout(feat.unsqueeze(1).expand(-1, T, -1).contiguous(), 0.1, self.training)

    def _head_parts(self, dec_in, x, feat):
        dec_out = ops.layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        return dec_out, torch.cat([dec_in, dec_out, feat], dim=-1)

    def _head(self, dec_in, x, ctxs, copies, feat, source_map):
        dec_out, gen_in = self._head_parts(dec_in, x, feat)
        gen = self._generate(gen_in, self.gen[1].p)
        d1, d2 = self._mix(dec_out, ctxs, gen, copies, source_map)
        return dec_out, gen, ((d1, d2) if self.training else ops.add(d1, d2))

    def _step(self, dec_ids, mems, valid, weights, answer_rep, source_map, cache=None):
        feat = self._feature(answer_rep, dec_ids.size(1))
        dec_in, x, ctxs, copies = self._run_prefix(dec_ids, mems, valid, weights, feat, cache)
        dec_out, gen, (d1, d2) = self._head(dec_in, x, ctxs, copies, feat, source_map)
        return dec_out, gen, d1, d2

    def forward(self, encode_memories, BOS, UNK, source_map, groundtruth_index=None, additional_decoder_feature=None,
                encode_weights=None, encode_masks=None, init_decoder_state=None, max_target_length=None):
        B = source_map.size(0)
        source_map = self._sorted(source_map)
        mems, valid, weights = self._prepare(encode_memories, encode_masks, encode_weights, B)
        if max_target_length is None:
            max_target_length = groundtruth_index.size(1)
        bos = self._bos(B, BOS, mems[0].device)
        if self.training and groundtruth_index is not None:
            dec_ids = torch.cat([bos, groundtruth_index[:, :-1]], dim=-1)
            dec_out, gen, d1, d2 = self._step(dec_ids, mems, valid, weights, additional_decoder_feature, source_map)
            return dec_out, gen, (d1, d2), groundtruth_index
        if self.training:
            return None
        return self._greedy(mems, valid, weights, source_map, BOS, max_target_length,
                            feature_of=lambda T: self._feature(additional_decoder_feature, T))


class RelevantPassageSelection(nn.Module):
    """Interaction -> 3 query / 5 passage TransformerBlocks -> Linear(H, 1) on [CLS]  (reference :127-163)."""

    def __init__(self, hidden_size, num_heads, query_encoder, passage_encoder):
        super().__init__()
        self.hidden_size = hidden_size
        self.query_encoder = query_encoder
        self.passage_encoder = passage_encoder
        self.num_heads = num_heads
        self.interaction = Interaction(hidden_size)
        self.query_blocks = block_stack(num_heads, hidden_size, 2)
        self.passage_blocks = block_stack(num_heads, hidden_size, 4)
        self.scorer = nn.Linear(hidden_size, 1)

    def action(self, query, passage, encode_query, encode_passage):
        eq, ep = encode_query[0][:, :, -1], encode_passage[0][:, :, -1]
        passage_mask, query_mask = passage.ne(0), query.ne(0)
        g_pq, g_qp = self.interaction(eq, ep, query_mask, passage_mask)
        query_reps, passage_reps = run_block_pair(self.query_blocks, g_pq, query_mask, self.passage_blocks, g_qp, passage_mask)
        cls = passage_reps[:, :, 0].contiguous()
        score = ops.linear(cls, self.scorer.weight, self.scorer.bias, out_dtype=torch.float32).squeeze(-1)
        return score, (query_reps, query_reps[:, :, 0]), (passage_reps, passage_reps[:, :, 0])


class SupportingTokenIdentification(nn.Module):
    """Second Interaction on the selection-stage reps -> 2 + 3 blocks -> per-token logit; reps refined by
    LN(stage1 + stage2)  (reference :165-212)."""

    def __init__(self, max_span_size, hidden_size, num_heads, query_encoder, passage_encoder, passage_selection):
        super().__init__()
        self.hidden_size = hidden_size
        self.num_heads = num_heads
        self.query_encoder = query_encoder
        self.passage_encoder = passage_encoder
        self.max_span_size = max_span_size
        self.passage_selection = passage_selection
        self.interaction = Interaction(hidden_size)
        self.query_blocks = block_stack(num_heads, hidden_size, 1)
        self.passage_blocks = block_stack(num_heads, hidden_size, 2)
        self.norm1 = nn.LayerNorm(hidden_size)
        self.norm2 = nn.LayerNorm(hidden_size)
        self.scorer = nn.Linear(hidden_size, 1)

    def action(self, query, passage, encode_query, encode_passage, passage_selection_result):
        passage_mask, query_mask = passage.ne(0), query.ne(0)
        _, query_rep, passage_rep = passage_selection_result
        g_pq, g_qp = self.interaction(query_rep[0], passage_rep[0], query_mask, passage_mask)
        query_reps, passage_reps = run_block_pair(self.query_blocks, g_pq, query_mask, self.passage_blocks, g_qp, passage_mask)
        token_score = ops.linear(passage_reps, self.scorer.weight, self.scorer.bias, out_dtype=torch.float32).squeeze(-1)
        token_score = token_score.masked_fill(~passage_mask, -1e6).clamp(min=-1e6, max=1e6)
        query_reps = ops.layer_norm(query_rep[0], self.norm1.weight, self.norm1.bias, self.norm1.eps, add=query_reps)
        passage_reps = ops.layer_norm(passage_rep[0], self.norm2.weight, self.norm2.bias, self.norm2.eps, add=passage_reps)
        return token_score, (query_reps, query_reps[:, :, 0]), (passage_reps, passage_reps[:, :, 0])


class ResponseGeneration(nn.Module):
    """Priors over the passage tokens + answer representation + decoder call (reference :214-253)."""

    def __init__(self, BOS, UNK, vocab_size, hidden_size, num_heads, query_encoder, passage_encoder, passage_selection,
                 span_extraction, decoder):
        super().__init__()
        self.hidden_size = hidden_size
        self.vocab_size = vocab_size
        self.num_heads = num_heads
        self.query_encoder = query_encoder
        self.passage_encoder = passage_encoder
        self.passage_selection = passage_selection
        self.span_extraction = span_extraction
        self.BOS = BOS
        self.UNK = UNK
        self.decoder = decoder

    def action(self, query, passage, source_map, encode_query, encode_passage, passage_selection_result,
               span_extraction_result, output=None, max_target_length=None):
        B = query.size(0)
        passage_score = passage_selection_result[0]
        token_score, query_rep, passage_rep = span_extraction_result
        H = passage_rep[0].size(-1)
        # sigma(passage) * sigma(token), normalised over all P*Lp tokens (:239-241): [B, P*Lp] f32 scalars (glue)
        prior = (torch.sigmoid(passage_score).unsqueeze(-1) * torch.sigmoid(token_score)).reshape(B, -1)
        prior = prior / (1e-8 + prior.sum(dim=-1, keepdim=True))
        mem = passage_rep[0].reshape(B, -1, H)
        answer_rep = ops.bmm(ops.cast_to(prior.unsqueeze(1), mem.dtype), mem, b_is_kn=True).squeeze(1)  # prior @ reps (:242)
        prior_p = prior.reshape_as(token_score)
        prior_q = torch.ones(B, 1, query_rep[0].size(2), device=prior.device)
        return self.decoder([query_rep[0], passage_rep[0]], self.BOS, self.UNK, source_map,
                            additional_decoder_feature=answer_rep, groundtruth_index=output,
                            max_target_length=max_target_length, encode_masks=[query.ne(0), passage.ne(0)],
                            encode_weights=[prior_q, prior_p])


class CaSE(nn.Module):
    def __init__(self, max_span_size, max_target_length, id2vocab, vocab2id, hidden_size, enc_layers=3, dec_layers=4, heads=8, early_stop=False):
        super().__init__()
        V = len(vocab2id)
        self.UNK = vocab2id[UNK_WORD]
        self.max_target_length = max_target_length
        self.query_encoder = TransformerSeqEncoder(enc_layers, heads, V, hidden_size)
        self.passage_encoder = self.query_encoder
        self.passage_selection = RelevantPassageSelection(hidden_size, heads, self.query_encoder, self.passage_encoder)
        self.span_extraction = SupportingTokenIdentification(max_span_size, hidden_size, heads, self.query_encoder,
                                                             self.passage_encoder, self.passage_selection)
        self.response_generation = ResponseGeneration(vocab2id[BOS_WORD], vocab2id[UNK_WORD], V, hidden_size, heads,
                                                      self.query_encoder, self.passage_encoder, self.passage_selection,
                                                      self.span_extraction,
                                                      CaSETransformerSeqDecoder(2, dec_layers, heads, V, hidden_size))
        self.id2vocab = id2vocab
        self.vocab_size = len(id2vocab)
        self.vocab2id = vocab2id
        if early_stop:  # greedy decoding ends once every answer of the batch has produced EOS (off = the reference's fixed T steps)
            self.response_generation.decoder.eos_id = vocab2id[EOS_WORD]

    def to_sentence(self, data, batch_indices):
        return to_sentence(batch_indices, self.id2vocab)

    def _encode_select_extract(self, data):
        if self.query_encoder is self.passage_encoder:  # one shared encoder (reference :262-263): both inputs in one pass
            eq, ep = self.query_encoder.forward_many([data['query'], data['passage']])
        else:
            eq, ep = self.query_encoder(data['query']), self.passage_encoder(data['passage'])
        ps = self.passage_selection.action(data['query'], data['passage'], encode_query=eq, encode_passage=ep)
        se = self.span_extraction.action(data['query'], data['passage'], encode_query=eq, encode_passage=ep,
                                         passage_selection_result=ps)
        return eq, ep, ps, se

    def do_train(self, data):
        eq, ep, ps, se = self._encode_select_extract(data)
        loss_ps = passage_bce(ps[0], data['passage_label'])
        valid = data['passage'].ne(0).float()
        bce = torch.nn.functional.binary_cross_entropy_with_logits(se[0], data['token_label'].detach(), reduction='none')
        loss_se = (valid * bce * data['token_weight'].detach()).sum() / valid.sum()
        rg = self.response_generation.action(data['query'], data['passage'], data['source_map'], encode_query=eq,
                                             encode_passage=ep, passage_selection_result=ps, span_extraction_result=se,
                                             output=data['response'])
        dist1, dist2 = rg[2]
        return [loss_ps, loss_se, generation_nll(ops.add(dist1, dist2), data['response'])]

    def do_test(self, data):
        eq, ep, ps, se = self._encode_select_extract(data)
        rg = self.response_generation.action(data['query'], data['passage'], data['source_map'], encode_query=eq,
                                             encode_passage=ep, passage_selection_result=ps, span_extraction_result=se,
                                             output=None, max_target_length=self.max_target_length)
        return {'answer': rg[3], 'rank': ps[0]}

    do_infer = do_test  # BASELINE.json's wording

    def forward(self, data, method='mle_train'):
        # the reference expands data['source_map'] into a dense one-hot here (Utils.build_map, 15 GB at cfg 2);
        # the ids themselves feed the pointer scatter kernel instead
        if method == 'train':
            return self.do_train(data)
        elif method == 'test':
            return self.do_test(data)
