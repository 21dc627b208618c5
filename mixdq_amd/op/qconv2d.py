"""qconv2d wrapper with the reference's positional order (mixdq_extension/op/qconv2d.py:4-22)."""
from mixdq_amd import _C


def qconv2d(input_int, weight_int, weight_scale, input_scale, input_zp, scale,
            weight_sum_by_input_channels, bias0, bias=None, stride=1, padding=0):
    dilation = 1
    return _C.qconv2d_w8_a8_ohalf(input_int, weight_int, weight_scale, input_scale, input_zp,
                                  scale, weight_sum_by_input_channels, bias0, bias, stride,
                                  padding, dilation)
