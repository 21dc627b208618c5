from mixdq_amd.nn.Linear import QuantizedLinear  # noqa: F401
from mixdq_amd.nn.Conv2d import QuantizedConv2d  # noqa: F401
from mixdq_amd.nn.glue import (HipAttnProcessor, HipGroupNorm, HipLayerNorm, HipSiLU,  # noqa: F401
                                swap_glue_modules, unswap_glue_modules)
