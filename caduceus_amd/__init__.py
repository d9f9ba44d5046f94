"""caduceus_amd: MI355X-native (gfx950) engine for the Caduceus forward/backward hot path.

Drop-in for the reference's `caduceus` package (/root/reference/caduceus/__init__.py:5-7): same class names, config
fields, module tree and state-dict keys; arithmetic runs on hand-written HIP kernels (libcaduceus_hip.so, C-ABI in
include/caduceus_hip.h).  There is no CPU fallback: without the built library / a GPU the ops raise.
"""
from .configuration_caduceus import CaduceusConfig
from .modeling_caduceus import (BiMambaWrapper, Caduceus, CaduceusForMaskedLM, CaduceusForSequenceClassification,
                                CaduceusMixerModel, create_block)
from .modeling_rcps import RCPSAddNormWrapper, RCPSEmbedding, RCPSLMHead, RCPSMambaBlock, RCPSWrapper
from .tokenization_caduceus import CaduceusTokenizer
from .downstream import DNAEmbeddingModelCaduceus, SequenceDecoder

__all__ = ["DNAEmbeddingModelCaduceus", "SequenceDecoder", "CaduceusConfig", "Caduceus", "CaduceusForMaskedLM", "CaduceusForSequenceClassification",
           "CaduceusTokenizer", "CaduceusMixerModel", "BiMambaWrapper", "create_block", "RCPSEmbedding", "RCPSWrapper",
           "RCPSAddNormWrapper", "RCPSMambaBlock", "RCPSLMHead", "register_auto_classes"]


def register_auto_classes():
    """Make `AutoConfig/AutoModel/AutoModelForMaskedLM.from_pretrained(<local dir>)` resolve model_type "caduceus"
    to this package offline (the reference relies on hub remote code, README.md:26-48)."""
    from transformers import AutoConfig, AutoModel, AutoModelForMaskedLM, AutoModelForSequenceClassification
    try:
        AutoConfig.register("caduceus", CaduceusConfig)
    except ValueError:
        pass
    for auto, cls in ((AutoModel, Caduceus), (AutoModelForMaskedLM, CaduceusForMaskedLM),
                      (AutoModelForSequenceClassification, CaduceusForSequenceClassification)):
        try:
            auto.register(CaduceusConfig, cls)
        except ValueError:
            pass
