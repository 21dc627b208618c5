"""Operator aliases, as mixdq_extension/op/qlinear.py:5-6."""
from mixdq_amd import _C

qlinear = _C.qlinear_w8_a8_ohalf
qlinear_ref = _C.qlinear_fp_reference
