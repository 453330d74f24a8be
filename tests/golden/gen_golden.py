"""Generate the golden fixtures by importing the reference itself (build container only).

    python tests/golden/gen_golden.py            # writes tests/golden/<case>.npz

No-ops (exit 0 with a message) when /root/reference is absent -- the GPU box never has it; only the
.npz data (inputs + expected outputs) travels.  Nothing from the reference is copied: this script
imports it in place, with three third-party modules stubbed (bcolz / nltk / transformers are
imported at the top of common/Utils.py but unused on the hot path) and dropout patched
to identity (bit-parity with torch's dropout RNG is impossible; the reference hard-codes
F.dropout(p=0.1) at TransformerBlock.py:27-28 and CaSE/Model.py:69,98).
"""
import importlib.machinery
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def reference_namespace():
    sys.dont_write_bytecode = True  # the reference mount is read-only
    sys.path.insert(0, REF)
    for name in ("bcolz", "nltk", "transformers", "transformers.optimization"):
        mod = types.ModuleType(name)
        mod.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
        mod.__all__ = []
        sys.modules[name] = mod
    # under torch 1.4 common/Utils.py got ``torch`` / ``math`` through star-imports that modern torch no longer leaks
    tr = sys.modules["transformers"]
    tr.torch, tr.math, tr.__all__ = torch, math, ["torch", "math"]
    torch.nn.functional.dropout = lambda x, p=0.5, training=True, inplace=False: x

    import CaSE.Model as case_model
    import Masque.Model as masque_model
    import common.BilinearAttention as ba
    import common.Highway as hw
    import common.Interaction as inter
    import common.PositionalEmbedding as pe
    import common.TransformerBlock as tb
    import common.TransformerDecoder as td
    import common.TransformerEncoder as te
    import common.TransformerSeqEncoderDecoder as sed
    import common.CumulativeTrainer as trainer
    import common.Utils as utils
    import evaluation.Eval_Rouge as eval_rouge
    import evaluation.Rouge as rouge

    ns = types.SimpleNamespace(
        PositionalEmbedding=pe.PositionalEmbedding, TransformerEncoderLayer=te.TransformerEncoderLayer,
        TransformerEncoder=te.TransformerEncoder, TransformerDecoderLayer=td.TransformerDecoderLayer,
        GenericTransformerDecoderLayer=td.GenericTransformerDecoderLayer, TransformerDecoder=td.TransformerDecoder,
        TransformerBlock=tb.TransformerBlock, BilinearAttention=ba.BilinearAttention, Interaction=inter.Interaction,
        Highway=hw.Highway, TransformerSeqEncoder=sed.TransformerSeqEncoder, TransformerSeqDecoder=sed.TransformerSeqDecoder,
        CaSE=case_model.CaSE, Masque=masque_model.Masque,
        generate_square_subsequent_mask=utils.generate_square_subsequent_mask, build_map=utils.build_map,
        universal_sentence_embedding=utils.universal_sentence_embedding, topk=utils.topk,
        CumulativeTrainer=trainer.CumulativeTrainer, lr_schedule=_lr_schedule(), to_sentence=utils.to_sentence,
        remove_duplicate=utils.remove_duplicate,
        rouge_l=lambda hyp, ref: rouge.rouge_l_sentence_level([hyp], [ref]),
        eval_rouge_l=lambda run, ref: eval_rouge.eval_rouge(run, ref)["ROUGE_L_F1"])
    return ns


def _lr_schedule():
    """The reference takes its schedule from transformers==2.1.1 (CaSE/Run.py:28), stubbed here: the trainer fixture runs the
    reference's loop with this package's restatement of that schedule (schedule parity is unpinned)."""
    from case_rg_amd.common.schedule import get_cosine_with_hard_restarts_schedule_with_warmup
    return get_cosine_with_hard_restarts_schedule_with_warmup


def to_numpy(rec):
    out = {}
    for k, v in rec.items():
        v = v.detach().cpu()
        out[k] = v.numpy()
    return out


def main():
    if not os.path.isdir(REF):
        print("gen_golden: %s not present; fixtures are generated in the build container only" % REF)
        return 0
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import cases

    ns = reference_namespace()
    torch.manual_seed(0)
    only = set(sys.argv[1:])
    for name, fn in cases.CASES.items():
        if only and name not in only:
            continue
        rec = to_numpy(fn(ns, torch.device("cpu")))
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **rec)
        print("%-24s %3d arrays %8.1f KB" % (name, len(rec), os.path.getsize(path) / 1024))
    return 0


if __name__ == "__main__":
    sys.exit(main())
