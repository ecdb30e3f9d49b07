"""vyomai_amd -- MI355X-native implementation of VyomAI's transformer hot path.

Same public names as the reference package (VyomAI/__init__.py:1-12) for everything on that
path; the math runs in hand-written gfx950 HIP kernels behind the C ABI in include/vyom_hip.h.
"""
from .utils import EncoderConfig  # noqa: F401
from .layers.kv_cache import DynamicCache, StaticCache, StaticCacheOne, DynamicCacheOne  # noqa: F401
from .models.encoder import EncoderModel, EncoderForMaskedLM  # noqa: F401
from .models.decoder import DecoderModel  # noqa: F401
from .models.vision_encoder import Vit  # noqa: F401
from .models.multimodel import VisionLanguageModel  # noqa: F401
from .models.encoder_decoder import EncoderDecoderModel  # noqa: F401
from .generation_utils import generate, generate_multimodel, generate_seq2seq  # noqa: F401
from .logits_processors import (LogitsProcessor, GreedyProcessor, MultinomialProcessor, TopKProcessor,  # noqa: F401
                                NucleusProcessor, TopKNucleusProcessor)
from .speculative_decoding import speculative_generate  # noqa: F401
