"""CPU oracle for the CaSE_RG encoder-decoder hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch fp32 CPU restatement of the
reference algorithm (file:line citations in every docstring).  It is imported only by
``tests/``, by ``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of ``bench.py`` --
always as the *checker* (or the timed CPU baseline), never as the thing shipped.  The product
package ``case_rg_amd`` never imports it and has no CPU fallback.

Parity pin: every module here is checked against golden vectors produced by importing the
reference itself in the build container (``tests/golden/gen_golden.py`` -> ``tests/golden/*.npz``,
checked by ``tests/test_oracle_vs_golden.py``).
"""
from .blocks import (  # noqa: F401
    sinusoid_table, PositionalEmbedding, MultiheadAttention, TransformerEncoderLayer,
    TransformerEncoder, TransformerDecoderLayer, GenericTransformerDecoderLayer,
    TransformerDecoder, TransformerBlock, BilinearAttention, Interaction, Highway,
    causal_additive_mask, masked_mean, one_hot_map, generate_square_subsequent_mask, build_map,
    universal_sentence_embedding, topk, to_sentence, remove_duplicate,
)
from .models import (  # noqa: F401
    TransformerSeqEncoder, TransformerSeqDecoder, CaSETransformerSeqDecoder,
    MasqueTransformerSeqDecoder, CaSE, Masque, SPECIALS,
)
from .rouge import rouge_l, eval_rouge_l  # noqa: F401,E402
