"""Sequence encoder and pointer-generator decoders on the HIP path.

Reference: common/TransformerSeqEncoderDecoder.py:14-45 (encoder), :47-150 (generic decoder);
CaSE/Model.py:13-125 and Masque/Model.py:13-119 hold the two task-specific decoder variants, which share
``PointerDecoderCore`` here.

Differences from the reference that do not change results:
  * the copy distribution is a scatter-add over the source *ids* (kernel K11) instead of a dense bmm with a
    [B, S, V] one-hot map (15-40 GB at the BASELINE sizes); a dense map is still accepted for API compatibility;
  * greedy decoding keeps the reference's step semantics (argmax of the last position, lowest index on ties, fixed
    T steps) but projects each memory's K/V and additive-attention keys once instead of once per step.
"""
import os

import torch
import torch.nn as nn

from .. import config, ops
from .BilinearAttention import BilinearAttention
from .PositionalEmbedding import PositionalEmbedding
from .TransformerDecoder import TransformerDecoder, TransformerDecoderLayer
from .TransformerEncoder import TransformerEncoder, TransformerEncoderLayer
from .Utils import generate_square_subsequent_mask

_generate_square_subsequent_mask = generate_square_subsequent_mask
MERGED_ENCODE = os.environ.get("CASE_MERGED_ENCODE", "1") != "0"  # A/B switch of TransformerSeqEncoder.forward_many
QUERY_SPLIT = os.environ.get("CASE_QUERY_SPLIT", "1") != "0"  # A/B switch: the greedy step projects x_t alone for the additive-attention query


def _embedding(vocab, width, max_len=1000, emb_matrix=None):
    """Embedding + position pair.  ``emb_matrix`` [V, H]: a pre-trained table, loaded and frozen as the reference's
    ``create_emb_layer`` does (common/Utils.py:250-256: ``non_trainable=True``)."""
    if emb_matrix is not None:
        vocab, width = emb_matrix.shape
    table = nn.Embedding(vocab, width, padding_idx=0)
    if emb_matrix is not None:
        table.load_state_dict({"weight": emb_matrix})
        table.weight.requires_grad = False
    return nn.Sequential(table, PositionalEmbedding(width, dropout=0.1, max_len=max_len))


def _embed(seq, ids, training):
    """Fused gather * sqrt(H) + position (+dropout) through the nn.Sequential(Embedding, PositionalEmbedding) pair."""
    table, pos = seq[0].weight, seq[1]
    if ids.shape[-1] > pos.pe.size(0):
        raise RuntimeError("sequence length %d exceeds max_len %d" % (ids.shape[-1], pos.pe.size(0)))
    return ops.embed_pos(ids, table, pos.pe, p_drop=config.drop_p(pos.p, training))


class TransformerSeqEncoder(nn.Module):
    def __init__(self, num_layers, num_heads, src_vocab_size, hidden_size, emb_matrix=None, norm=None):
        super().__init__()
        self.num_layers, self.num_heads = num_layers, num_heads
        self.embedding = _embedding(src_vocab_size, hidden_size, emb_matrix=emb_matrix)
        layer = TransformerEncoderLayer(hidden_size, nhead=num_heads, dim_feedforward=hidden_size, dropout=0.1, activation='gelu')
        self.enc = TransformerEncoder(layer, num_layers=num_layers, norm=norm)

    def forward(self, batch_numseq_seqlen):
        """ids [B, N, L] -> (out [B, N, 1, L, H], state [B, N, 1, H])."""
        B, N, L = batch_numseq_seqlen.shape
        ids = batch_numseq_seqlen.reshape(B * N, L)
        valid = ids.ne(0)
        x = _embed(self.embedding, ids, self.training)
        y = self.enc.forward_batch_first(x, valid)
        state = ops.masked_mean(y, valid)
        return y.reshape(B, N, L, -1).unsqueeze(2), state.reshape(B, N, -1).unsqueeze(2)


    def forward_many(self, id_tensors):
        """Several inputs through the SAME encoder in one pass (the reference's models share one encoder between query and passages:
        CaSE/Model.py:262-263, Masque/Model.py:207-208): embeddings of all inputs back to back in one [rows, H] buffer, every
        row-local op of every layer launched once, the attention core once per input.  Returns what ``forward`` returns, per input;
        the training path's ~120 launches on the 2048 query rows ride along with the 122 880 passage rows, and the shared
        parameters receive ONE gradient instead of two that autograd adds.  Falls back to separate passes where the grouped attention
        does not apply (f32 parity mode, the inference chain, unfused mode)."""
        dt = config.compute_dtype()
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if (not MERGED_ENCODE or len(id_tensors) < 2 or not needs_grad or not self.enc.rows_supported(dt, needs_grad)
                or not all(t.is_cuda for t in id_tensors)):
            return [self.forward(t) for t in id_tensors]
        xs, groups, valids, shapes, r0 = [], [], [], [], 0
        for t in id_tensors:
            B, N, L = t.shape
            ids = t.reshape(B * N, L)
            xs.append(_embed(self.embedding, ids, self.training).reshape(B * N * L, -1))
            groups.append((r0, B * N, L))
            valids.append(ids.ne(0))
            shapes.append((B, N, L))
            r0 += B * N * L
        y = self.enc.forward_rows(torch.cat(xs, dim=0), groups, valids)
        outs = []
        for yi, valid, (B, N, L) in zip(ops.split_rows(y, groups), valids, shapes):
            state = ops.masked_mean(yi, valid)
            outs.append((yi.reshape(B, N, L, -1).unsqueeze(2), state.reshape(B, N, -1).unsqueeze(2)))
        return outs


class PointerDecoderCore(nn.Module):
    """Shared machinery of the three pointer-generator decoders."""

    def _build(self, num_memories, num_layers, nhead, vocab, H, query_width, emb_matrix=None, max_len=1000):
        self.tgt_vocab_size, self.num_layers, self.hidden_size = vocab, num_layers, H
        self.embedding = _embedding(vocab, H, max_len=max_len, emb_matrix=emb_matrix)
        layer = TransformerDecoderLayer(H, nhead=nhead, dim_feedforward=H, dropout=0.1, activation='gelu')
        self.decs = nn.ModuleList([TransformerDecoder(layer, num_layers=num_layers, norm=None) for _ in range(num_memories)])
        self.attns = nn.ModuleList([BilinearAttention(query_width, H, H) for _ in range(num_memories)])
        # EOS-aware early stop of greedy decoding.  None = the reference's behaviour: always max_target_length steps
        # (CaSE/Model.py:94).  With an id, a finished answer is continued with PAD and the loop ends once EVERY answer of the
        # batch has produced EOS -- checked every ``eos_check_every`` steps, so one host sync per 8 tokens, none per token.
        self.eos_id = None
        self.eos_check_every = 8
        self.last_greedy_steps = 0

    # ------------------------------------------------------------------------------------------
    def _prepare(self, encode_memories, encode_masks, encode_weights, batch_size):
        H = self.hidden_size
        mems = [m.reshape(batch_size, -1, H) for m in encode_memories]
        valid = [m.reshape(batch_size, -1).contiguous() for m in encode_masks]
        weights = None if encode_weights is None else [w.reshape(batch_size, -1) for w in encode_weights]
        return mems, valid, weights

    def _sorted(self, source_map):
        """The batch's source map with its (token, position) keys sorted on the device, once per forward: every
        pointer scatter of the batch -- one per training step, one per generated token -- then adds run by run without atomics."""
        if torch.is_tensor(source_map) and source_map.is_cuda and ops.SortedSource.fits(source_map, self.tgt_vocab_size):
            return ops.SortedSource(source_map, self.tgt_vocab_size)
        return source_map

    def _memory_cache(self, mems, absorb=False):
        """Step-invariant projections of the memories (cross-attention K/V per layer, additive-attention keys)."""
        out = []
        for i, m in enumerate(mems):
            fused = absorb and ops.pointer_decode_supported(m) and self.attns[i].hidden_size == m.shape[2]  # K22: e^{2 uh} instead of uh
            out.append(dict(kvs=self.decs[i].project_memory(m, absorb=absorb), uh=None if fused else self.attns[i].project_keys(m),
                            eu=self.attns[i].project_keys_exp(m) if fused else None))
        return out

    def _run_prefix(self, dec_ids, mems, valid, weights, feature, cache=None):
        """decs[0] -> attns[0] -> decs[1] -> attns[1] (a sequential chain, CaSE/Model.py:74-83)."""
        dec_in = _embed(self.embedding, dec_ids, self.training)
        tgt_valid = dec_ids.ne(0)
        x = dec_in
        ctxs, copies = [], []
        for i, mem in enumerate(mems):
            c = None if cache is None else cache[i]
            x = self.decs[i].forward_batch_first(x, mem, tgt_valid, valid[i], causal=True,
                                                 memory_kvs=None if c is None else c["kvs"])
            q = x if feature is None else torch.cat([x, feature], dim=-1)
            ctx, p = self.attns[i].attend(q, mem, mem, row_valid=tgt_valid, col_valid=valid[i],
                                          uh=None if c is None else c["uh"])
            ctxs.append(ctx)
            if weights is not None:
                p = weights[i].unsqueeze(1) * p
                p = p / (1e-8 + p.sum(dim=-1, keepdim=True))
            copies.append(p)
        return dec_in, x, ctxs, copies

    def _greedy(self, mems, valid, weights, source_map, BOS, max_target_length, feature_of=None):
        """KV-cached greedy decoding (K13).  Step semantics are the reference's (CaSE/Model.py:94-123): fixed number of steps,
        argmax of the newest position with the lowest index on ties, PAD tokens in the prefix masked as keys -- but each step
        computes ONE new position: self-attention K/V of earlier positions, the per-layer K/V projections of both memories and
        the additive-attention keys are cached, so a step streams the caches once instead of recomputing the prefix
        (O(T) instead of O(T^2) decoder work, no per-step projection of the 3840-token passage memory)."""
        B, dev = mems[0].shape[0], mems[0].device
        # the raw-pointer decode kernels (K21 / K22 / K23) have no autograd Function behind them: eval mode AND no_grad (the reference only
        # predicts under no_grad; a caller that differentiates an eval-mode greedy pass keeps the differentiable launches)
        inference = not self.training and not torch.is_grad_enabled()
        cache = self._memory_cache(mems, absorb=inference)
        self_kvs = [dec.new_self_cache(B, max_target_length, mems[0]) for dec in self.decs]
        hist_valid = torch.zeros(B, max_target_length, dtype=torch.bool, device=dev)
        table, pos = self.embedding[0].weight, self.embedding[1]
        if max_target_length > pos.pe.size(0):
            raise RuntimeError("max_target_length %d exceeds max_len %d" % (max_target_length, pos.pe.size(0)))
        ids = self._bos(B, BOS, dev)
        feat = None if feature_of is None else feature_of(1)
        # the feature half of the additive-attention query does not change over the steps: projected once per pass (BilinearAttention.split_query)
        splits = [self.attns[i].split_query(feat, self.hidden_size) if (QUERY_SPLIT and inference and feat is not None and cache[i]["eu"] is not None) else None
                  for i in range(len(mems))]
        picked = []
        finished = None if self.eos_id is None else torch.zeros(B, dtype=torch.bool, device=dev)
        capturing = torch.cuda.is_current_stream_capturing()  # a captured pass cannot branch on device data: fixed T steps
        fused_head = (inference and hasattr(self, "_head_parts") and ops.pointer_head_supported(source_map, self.tgt_vocab_size, len(mems)))
        for t in range(max_target_length):
            tok_valid = ids.ne(0)
            hist_valid[:, t] = tok_valid[:, 0]
            dec_in = ops.embed_pos(ids, table, pos.pe[t:t + 1])  # position t
            x = dec_in
            ctxs, copies = [], []
            for i, mem in enumerate(mems):
                x = self.decs[i].step(x, t, self_kvs[i], hist_valid, cache[i]["kvs"], valid[i])
                q = x if (feat is None or splits[i] is not None) else torch.cat([x, feat], dim=-1)
                if cache[i]["eu"] is not None:  # K22: scores, softmax, prior renormalisation and context in one launch
                    ctx, p = self.attns[i].attend_decode(q, mem, tok_valid, valid[i], cache[i]["eu"], None if weights is None else weights[i],
                                                         split=splits[i])
                else:
                    ctx, p = self.attns[i].attend(q, mem, mem, row_valid=tok_valid, col_valid=valid[i], uh=cache[i]["uh"])
                    if weights is not None:
                        p = weights[i].unsqueeze(1) * p
                        p = p / (1e-8 + p.sum(dim=-1, keepdim=True))
                ctxs.append(ctx)
                copies.append(p)
            if fused_head:  # K23: vocabulary softmax, mixing, pointer scatter and argmax in one launch
                dec_out, gen_in = self._head_parts(dec_in, x, feat)
                # the distributions leave the device for the LAST step only (what the caller gets back); with an EOS-aware early stop any
                # step may turn out to be the last one
                want = t == max_target_length - 1 or (finished is not None and not capturing)
                gen, dist, ids = self._head_decode(dec_out, gen_in, ctxs, copies, source_map, want)
            else:
                dec_out, gen, dist = self._head(dec_in, x, ctxs, copies, feat, source_map)
                ids = ops.row_argmax(dist[:, -1])[0].unsqueeze(1)
            if finished is not None:
                ids = ids.masked_fill(finished.unsqueeze(1), 0)  # PAD behind a finished answer (to_sentence stops at EOS anyway)
                finished = finished | ids[:, 0].eq(self.eos_id)
            picked.append(ids)
            if (finished is not None and not capturing and (t + 1) % self.eos_check_every == 0 and t + 1 < max_target_length
                    and bool(finished.all())):
                break
        self.last_greedy_steps = len(picked)
        answer = torch.cat(picked, dim=-1)
        if answer.size(1) < max_target_length:
            answer = torch.nn.functional.pad(answer, (0, max_target_length - answer.size(1)))
        return dec_out, gen, dist, answer

    def _head_decode(self, dec_out, gen_in, ctxs, copies, source_map, want_dists=True):
        """The head of one greedy step through K23 (ops.pointer_head_decode): the two generator Linears and the mixing Linear, then ONE
        launch for softmax over V, softmax over the mixing logits, p0 x gen + the pointer scatter, and the argmax."""
        B, V = dec_out.shape[0], self.tgt_vocab_size
        h = ops.linear(gen_in, self.gen[0].weight, self.gen[0].bias)
        logits = ops.linear(h, self.gen[-2].weight, None, out_dtype=torch.float32)
        parts = [dec_out] + ctxs
        if ops.linear_skinny_supported(parts, self.mix.weight):  # the 1 + nmem mixing logits straight from the three inputs: no concatenation, no N = 3 GEMM tile
            mix_logits = ops.linear_skinny(parts, self.mix.weight, self.mix.bias)
        else:
            mix_logits = ops.linear(torch.cat(parts, dim=-1), self.mix.weight, self.mix.bias, out_dtype=torch.float32)
        gen, dist, ids = ops.pointer_head_decode(logits.reshape(B, V), mix_logits.reshape(B, -1), source_map, [c.reshape(B, -1) for c in copies],
                                                 want_gen=want_dists, want_dist=want_dists)
        if not want_dists:
            return None, None, ids.unsqueeze(1)
        return gen.view(B, 1, V), dist.view(B, 1, V), ids.unsqueeze(1)

    def _generate(self, gen_in, hidden_drop):
        """gen = softmax(W_v (drop(W_h x + b)))  -- f32 logits and probabilities (K10)."""
        h = ops.linear(gen_in, self.gen[0].weight, self.gen[0].bias, p_drop=config.drop_p(hidden_drop, self.training))
        logits = ops.linear(h, self.gen[-2].weight, None, out_dtype=torch.float32)
        return ops.masked_softmax(logits)

    def _mix(self, dec_out, ctxs, gen, copies, source_map):
        """p = softmax(mix([dec_out, ctx_q, ctx_p])); dist1 = p0 * gen; dist2 = pointer mass scattered to the vocabulary."""
        mix_in = torch.cat([dec_out] + ctxs, dim=-1)
        pm = ops.masked_softmax(ops.linear(mix_in, self.mix.weight, self.mix.bias, out_dtype=torch.float32))
        dist1 = pm[:, :, 0:1] * gen
        ptr = torch.cat([pm[:, :, k + 1:k + 2] * c for k, c in enumerate(copies)], dim=-1)
        return dist1, self._scatter(ptr, source_map, gen.shape[-1])

    @staticmethod
    def _scatter(ptr, source_map, V):
        if isinstance(source_map, ops.SortedSource) or (source_map.dtype == torch.int64 and source_map.dim() == 2):
            return ops.copy_scatter(source_map, ptr, V)
        # dense [B, S, V] one-hot given by an API-compatible caller: recover the ids once, then scatter
        return ops.copy_scatter(source_map.argmax(dim=-1), ptr * source_map.sum(dim=-1).unsqueeze(1), V)

    @staticmethod
    def _bos(batch_size, BOS, device):
        return torch.full((batch_size, 1), BOS, dtype=torch.long, device=device)


class TransformerSeqDecoder(PointerDecoderCore):
    """Generic multi-memory decoder (reference: common/TransformerSeqEncoderDecoder.py:47-150)."""

    def __init__(self, num_memories, num_layers, nhead, tgt_vocab_size, hidden_size, emb_matrix=None):
        super().__init__()
        H = hidden_size
        # the reference's generic decoder builds its position table with max_len = 100 when a pre-trained embedding matrix is given
        # (common/TransformerSeqEncoderDecoder.py:57; 1000 otherwise, :55): ``pe`` is a persistent buffer, so a checkpoint of that
        # configuration only loads strictly into the same shape
        self._build(num_memories, num_layers, nhead, tgt_vocab_size, H, H, emb_matrix=emb_matrix, max_len=100 if emb_matrix is not None else 1000)
        self.norm = nn.LayerNorm(H)
        self.gen = nn.Sequential(nn.Linear(2 * H, H), nn.Linear(H, tgt_vocab_size, bias=False), nn.Softmax(dim=-1))
        self.mix = nn.Linear(H + num_memories * H, num_memories + 1)

    def _head_parts(self, dec_in, x, feat):
        dec_out = ops.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return dec_out, torch.cat([dec_in, dec_out], dim=-1)

    def _head(self, dec_in, x, ctxs, copies, feat, source_map):
        dec_out, gen_in = self._head_parts(dec_in, x, feat)
        gen = self._generate(gen_in, 0.0)
        d1, d2 = self._mix(dec_out, ctxs, gen, copies, source_map)
        return dec_out, gen, d1 + d2

    def _step(self, dec_ids, mems, valid, weights, source_map, cache=None):
        dec_in, x, ctxs, copies = self._run_prefix(dec_ids, mems, valid, weights, None, cache)
        return self._head(dec_in, x, ctxs, copies, None, source_map)

    def _source(self, source_maps):
        return torch.cat(source_maps, dim=-2 if source_maps[0].dim() == 3 else -1)

    def forward(self, encode_memories, BOS, UNK, source_maps, encode_masks=None, encode_weights=None,
                groundtruth_index=None, init_decoder_state=None, max_target_length=None):
        source_map = self._source(source_maps) if isinstance(source_maps, (list, tuple)) else source_maps
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
