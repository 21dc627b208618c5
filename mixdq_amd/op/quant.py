"""Operator aliases, as mixdq_extension/op/quant.py:4-5."""
from mixdq_amd import _C

quantize_per_tensor = _C.quantize_per_tensor_to_int8
quantize_per_tensor_vectorized = _C.quantize_per_tensor_to_int8_vectorized
