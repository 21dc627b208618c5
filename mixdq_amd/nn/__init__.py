from mixdq_amd.nn.Linear import QuantizedLinear  # noqa: F401
from mixdq_amd.nn.Conv2d import QuantizedConv2d  # noqa: F401
