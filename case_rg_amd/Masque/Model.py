"""Masque task model on the HIP path (reference: Masque/Model.py:13-286): CaSE minus supporting-token
identification; two losses (+ a passage-selection-only training mode)."""
import torch
import torch.nn as nn

from .. import ops
from ..common.Constants import BOS_WORD, EOS_WORD, UNK_WORD
from ..common.Interaction import Interaction
from ..common.TransformerSeqEncoderDecoder import PointerDecoderCore, TransformerSeqEncoder
from ..common.Utils import to_sentence
from ..common.heads import block_stack, generation_nll, passage_bce, run_block_pair


class MasqueTransformerSeqDecoder(PointerDecoderCore):
    """Reference :13-119: additive-attention query = decoder state (H wide), gen on cat[dec_in, dec_out], one norm,
    ``extend`` always returns the summed distribution."""

    def __init__(self, num_memories, num_layers, nhead, tgt_vocab_size, hidden_size, emb_matrix=None):
        super().__init__()
        H = hidden_size
        self._build(num_memories, num_layers, nhead, tgt_vocab_size, H, H, emb_matrix=emb_matrix)
        self.norm = nn.LayerNorm(H)
        self.gen = nn.Sequential(nn.Linear(2 * H, H), nn.Linear(H, tgt_vocab_size, bias=False), nn.Softmax(dim=-1))
        self.mix = nn.Linear(3 * H, num_memories + 1)

    def extend(self, dec_outputs, gen_outputs, memory_weights, source_map):
        H = self.hidden_size
        d1, d2 = self._mix(dec_outputs[..., :H], [dec_outputs[..., H:2 * H], dec_outputs[..., 2 * H:]], gen_outputs,
                           memory_weights, source_map)
        return d1 + d2

    def _head_parts(self, dec_in, x, feat):
        dec_out = ops.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return dec_out, torch.cat([dec_in, dec_out], dim=-1)

    def _head(self, dec_in, x, ctxs, copies, feat, source_map):
        dec_out, gen_in = self._head_parts(dec_in, x, feat)
        gen = self._generate(gen_in, 0.0)
        d1, d2 = self._mix(dec_out, ctxs, gen, copies, source_map)
        return dec_out, gen, ops.add(d1, d2)

    def _step(self, dec_ids, mems, valid, weights, source_map, cache=None):
        dec_in, x, ctxs, copies = self._run_prefix(dec_ids, mems, valid, weights, None, cache)
        return self._head(dec_in, x, ctxs, copies, None, source_map)

    def forward(self, encode_memories, BOS, UNK, source_map, encode_masks=None, encode_weights=None,
                groundtruth_index=None, init_decoder_state=None, max_target_length=None):
        B = source_map.size(0)
        source_map = self._sorted(source_map)
        mems, valid, weights = self._prepare(encode_memories, encode_masks, encode_weights, B)
        if max_target_length is None:
            max_target_length = groundtruth_index.size(1)
        bos = self._bos(B, BOS, mems[0].device)
        if self.training and groundtruth_index is not None:
            dec_ids = torch.cat([bos, groundtruth_index[:, :-1]], dim=-1)
            dec_out, gen, dist = self._step(dec_ids, mems, valid, weights, source_map)
            return dec_out, gen, dist, groundtruth_index
        if self.training:
            return None
        return self._greedy(mems, valid, weights, source_map, BOS, max_target_length)


class PassageSelection(nn.Module):
    """Reference :121-159 (same network as CaSE's selection stage; returns bare tensors)."""

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

    def action(self, query, passage, encode_query=None, encode_passage=None):
        if encode_query is None:
            encode_query = self.query_encoder(query)[0][:, :, -1]
        if encode_passage is None:
            encode_passage = self.passage_encoder(passage)[0][:, :, -1]
        passage_mask, query_mask = passage.ne(0), query.ne(0)
        g_pq, g_qp = self.interaction(encode_query, encode_passage, query_mask, passage_mask)
        query_reps, passage_reps = run_block_pair(self.query_blocks, g_pq, query_mask, self.passage_blocks, g_qp, passage_mask)
        cls = passage_reps[:, :, 0].contiguous()
        score = ops.linear(cls, self.scorer.weight, self.scorer.bias, out_dtype=torch.float32).squeeze(-1)
        return score, query_reps, passage_reps


class ResponseGeneration(nn.Module):
    """Reference :161-200: passage prior = sigma(passage score) broadcast over its tokens."""

    def __init__(self, BOS, UNK, vocab_size, hidden_size, num_heads, query_encoder, passage_encoder, passage_selection, decoder):
        super().__init__()
        self.hidden_size = hidden_size
        self.vocab_size = vocab_size
        self.num_heads = num_heads
        self.query_encoder = query_encoder
        self.passage_encoder = passage_encoder
        self.passage_selection = passage_selection
        self.BOS = BOS
        self.UNK = UNK
        self.decoder = decoder

    def action(self, query, passage, source_map, encode_query=None, encode_passage=None, passage_selection_result=None,
               output=None, max_target_length=None):
        if encode_query is None:
            encode_query = self.query_encoder(query)[0][:, :, -1]
        if encode_passage is None:
            encode_passage = self.passage_encoder(passage)[0][:, :, -1]
        if passage_selection_result is None:
            passage_selection_result = self.passage_selection.action(query, passage, encode_query=encode_query,
                                                                     encode_passage=encode_passage)
        passage_score, query_rep, passage_rep = passage_selection_result
        B = query.size(0)
        prior_q = torch.ones(B, 1, query_rep.size(2), device=passage_score.device)
        prior_p = torch.sigmoid(passage_score).unsqueeze(-1).expand(-1, -1, passage_rep.size(2))
        return self.decoder([query_rep, passage_rep], self.BOS, self.UNK, source_map, groundtruth_index=output,
                            max_target_length=max_target_length, encode_masks=[query.ne(0), passage.ne(0)],
                            encode_weights=[prior_q, prior_p])


class Masque(nn.Module):
    def __init__(self, max_target_length, id2vocab, vocab2id, hidden_size, enc_layers=3, dec_layers=4, heads=8, early_stop=False):
        super().__init__()
        V = len(vocab2id)
        self.UNK = vocab2id[UNK_WORD]
        self.max_target_length = max_target_length
        self.query_encoder = TransformerSeqEncoder(enc_layers, heads, V, hidden_size)
        self.passage_encoder = self.query_encoder
        self.passage_selection = PassageSelection(hidden_size, heads, self.query_encoder, self.passage_encoder)
        self.response_generation = ResponseGeneration(vocab2id[BOS_WORD], vocab2id[UNK_WORD], V, hidden_size, heads,
                                                      self.query_encoder, self.passage_encoder, self.passage_selection,
                                                      MasqueTransformerSeqDecoder(2, dec_layers, heads, V, hidden_size))
        self.id2vocab = id2vocab
        self.vocab_size = len(id2vocab)
        self.vocab2id = vocab2id
        if early_stop:  # greedy decoding ends once every answer of the batch has produced EOS (off = the reference's fixed T steps)
            self.response_generation.decoder.eos_id = vocab2id[EOS_WORD]

    def to_sentence(self, data, batch_indices):
        return to_sentence(batch_indices, self.id2vocab)

    def _encode_select(self, data):
        if self.query_encoder is self.passage_encoder:  # one shared encoder (reference :207-208): both inputs in one pass
            oq, op = self.query_encoder.forward_many([data['query'], data['passage']])
            eq, ep = oq[0][:, :, -1], op[0][:, :, -1]
        else:
            eq = self.query_encoder(data['query'])[0][:, :, -1]
            ep = self.passage_encoder(data['passage'])[0][:, :, -1]
        return eq, ep, self.passage_selection.action(data['query'], data['passage'], encode_query=eq, encode_passage=ep)

    def do_train(self, data):
        eq, ep, ps = self._encode_select(data)
        rg = self.response_generation.action(data['query'], data['passage'], data['source_map'], encode_query=eq,
                                             encode_passage=ep, passage_selection_result=ps, output=data['response'])
        return [0.25 * passage_bce(ps[0], data['passage_label']), generation_nll(rg[2], data['response'])]

    def do_ps_train(self, data):
        _, _, ps = self._encode_select(data)
        return [passage_bce(ps[0], data['passage_label'])]

    def do_test(self, data):
        eq, ep, ps = self._encode_select(data)
        rg = self.response_generation.action(data['query'], data['passage'], data['source_map'], encode_query=eq,
                                             encode_passage=ep, passage_selection_result=ps, output=None,
                                             max_target_length=self.max_target_length)
        return {'answer': rg[3], 'rank': ps[0]}

    do_infer = do_test

    def forward(self, data, method='mle_train'):
        if method == 'train':
            return self.do_train(data)
        elif method == 'ps_train':
            return self.do_ps_train(data)
        elif method == 'test':
            return self.do_test(data)
