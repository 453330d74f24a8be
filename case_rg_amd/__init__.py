"""case_rg_amd -- MI355X-native implementation of the CaSE_RG encoder-decoder hot path.

Python hosts the reference's ``nn.Module`` surface (``case_rg_amd.common.*``, ``case_rg_amd.CaSE.Model``,
``case_rg_amd.Masque.Model``); all activation arithmetic runs in hand-written HIP kernels of
``csrc/libcase_hip.so`` behind the C ABI of ``include/case_hip.h``.  Importing this package loads that library
and fails loudly when it is missing -- there is no CPU or eager fallback.
"""
import sys as _sys
import types as _types

from . import _abi, config, ops  # noqa: F401  (loads libcase_hip.so)
from .config import compute_dtype, manual_seed, set_compute_dtype, set_dropout  # noqa: F401

__version__ = "0.1.0"


def namespace():
    """Reference-named classes and helpers in one namespace (what the shared parity cases consume)."""
    from .CaSE.Model import CaSE
    from .Masque.Model import Masque
    from .common import (BilinearAttention, Highway, Interaction, PositionalEmbedding, TransformerBlock, TransformerDecoder,
                         TransformerEncoder, TransformerSeqEncoderDecoder, Utils)
    return _types.SimpleNamespace(
        PositionalEmbedding=PositionalEmbedding.PositionalEmbedding,
        TransformerEncoderLayer=TransformerEncoder.TransformerEncoderLayer, TransformerEncoder=TransformerEncoder.TransformerEncoder,
        TransformerDecoderLayer=TransformerDecoder.TransformerDecoderLayer,
        GenericTransformerDecoderLayer=TransformerDecoder.GenericTransformerDecoderLayer,
        TransformerDecoder=TransformerDecoder.TransformerDecoder, TransformerBlock=TransformerBlock.TransformerBlock,
        BilinearAttention=BilinearAttention.BilinearAttention, Interaction=Interaction.Interaction, Highway=Highway.Highway,
        TransformerSeqEncoder=TransformerSeqEncoderDecoder.TransformerSeqEncoder,
        TransformerSeqDecoder=TransformerSeqEncoderDecoder.TransformerSeqDecoder, CaSE=CaSE, Masque=Masque,
        generate_square_subsequent_mask=Utils.generate_square_subsequent_mask, build_map=Utils.build_map,
        universal_sentence_embedding=Utils.universal_sentence_embedding, topk=Utils.topk)


def install_dropin():
    """Alias this package's ``common`` / ``CaSE`` / ``Masque`` as top-level modules so the reference's launch
    scripts (``from CaSE.Model import *``, ``from common.CumulativeTrainer import *``) resolve to the HIP path."""
    import importlib
    import pkgutil
    for name in ("common", "CaSE", "Masque"):
        pkg = importlib.import_module(__name__ + "." + name)
        _sys.modules[name] = pkg
        for info in pkgutil.iter_modules(pkg.__path__):
            sub = importlib.import_module("%s.%s.%s" % (__name__, name, info.name))
            _sys.modules["%s.%s" % (name, info.name)] = sub
