"""The module-swap surface and run harness: counterpart of kernels/quantize_sdxl.py.

    register_qconfig_from_input_files(unet, args, bos, bos_dict)    quantize_sdxl.py:39-139
    convert_to_quantized(unet, ckpt)                                 quantize_sdxl.py:142-150
    quantize_unet(unet, args, ckpt, bos, bos_dict)                   quantize_sdxl.py:154-156
    hip_graph_opt(unet)   (reference name cuda_graph_opt kept as alias)  quantize_sdxl.py:184-286
    example_inputs(...)                                              quantize_sdxl.py:350-373

`unet` is any nn.Module whose Linear/Conv2d sub-module names match the yaml keys: a diffusers
UNet2DConditionModel (where diffusers exists) or mixdq_amd.unet.SDXLUNet (this repo).
"""
from __future__ import annotations

import functools
import threading

import torch
import torch.nn as nn
from torch.ao.quantization import PlaceholderObserver, QConfig

BW_TO_DTYPE = {
    8: torch.qint8,
    4: torch.quint4x2,
    2: torch.quint4x2,   # 2-bit is not supported by the reference either: treated as 4
}


def _strip_model_prefix(name: str) -> str:
    """yaml keys are 'model.<unet module path>' (quantize_sdxl.py:56-60)."""
    if "model." in name:
        return name[name.index("model.") + 6:]
    return name


def load_bitwidth_config(path_or_dict):
    if isinstance(path_or_dict, dict):
        raw = path_or_dict
    else:
        import yaml
        with open(path_or_dict, "r") as f:
            raw = yaml.safe_load(f)
    return {_strip_model_prefix(k): int(v) for k, v in raw.items()}


def register_qconfig_from_input_files(unet, args, bos, bos_dict):
    """Attach .qconfig / .module_name / .w_bit / .a_bit (and .bos, .bos_pre_computed on the
    cross-attention to_k / to_v) to every layer named in the weight and activation yamls."""
    w_bits = load_bitwidth_config(args.w_config)
    pending = dict(w_bits)
    for name, mod in unet.named_modules():
        if name not in w_bits:
            continue
        assert not hasattr(mod, "qconfig")
        bits = w_bits[name]
        mod.qconfig = QConfig(
            activation=PlaceholderObserver.with_args(dtype=torch.float16),
            weight=PlaceholderObserver.with_args(dtype=BW_TO_DTYPE[bits]))
        mod.module_name = name
        mod.w_bit = bits
        if "attn2" in name and ("to_k" in name or "to_v" in name):
            mod.bos = bos
            mod.bos_pre_computed = bos_dict[name]
        del pending[name]
    if pending:
        for name in pending:
            print(f"{name} not found in UNet!")
        raise RuntimeError("Not all keys in weight yaml map to a module in UNet.")

    if getattr(args, "a_config", None) is None:
        return
    a_bits = load_bitwidth_config(args.a_config)
    pending = dict(a_bits)
    for name, mod in unet.named_modules():
        if name not in a_bits:
            continue
        act = PlaceholderObserver.with_args(dtype=BW_TO_DTYPE[a_bits[name]])
        if getattr(mod, "qconfig", None):
            assert isinstance(mod.qconfig, QConfig)
            mod.qconfig = QConfig(weight=mod.qconfig.weight, activation=act)
        else:
            mod.qconfig = QConfig(activation=act,
                                  weight=PlaceholderObserver.with_args(dtype=torch.float16))
        mod.a_bit = a_bits[name]
        del pending[name]
    if pending:
        for name in pending:
            print(f"{name} not found in UNet!")
        raise RuntimeError("Not all keys in act yaml map to a module in UNet.")


def convert_to_quantized(unet, ckpt):
    from mixdq_amd.nn import QuantizedConv2d, QuantizedLinear
    from mixdq_amd.quantize import convert
    convert(unet, mapping={nn.Linear: QuantizedLinear, nn.Conv2d: QuantizedConv2d},
            inplace=True, ckpt=ckpt)


def quantize_unet(unet, args, ckpt, bos, bos_dict, w4_kernel=False):
    """`w4_kernel=True` (not in the reference): 4-/2-bit weight layers run the packed-W4 INT8
    kernels instead of falling back to FP16 (mixdq_amd.nn.QuantizedLinear.w4_kernel)."""
    register_qconfig_from_input_files(unet, args, bos=bos, bos_dict=bos_dict)
    if w4_kernel:
        for mod in unet.modules():
            if getattr(mod, "w_bit", 8) in (2, 4):
                mod.w4_kernel = True
    convert_to_quantized(unet, ckpt)


# ---------------------------------------------------------------------------------------------
# hipGraph capture of the forward (torch.cuda.CUDAGraph is hipGraph on ROCm)
# ---------------------------------------------------------------------------------------------
def _arg_key(arg):
    if isinstance(arg, torch.Tensor):
        one = arg.item() if arg.device.type == "cpu" and arg.numel() == 1 else None
        return (arg.device.type, arg.device.index, arg.dtype, tuple(arg.shape), one)
    if isinstance(arg, (str, int, float, bytes, bool)):
        return arg
    if isinstance(arg, (tuple, list)):
        return tuple(_arg_key(a) for a in arg)
    if isinstance(arg, dict):
        return tuple(sorted(((_arg_key(k), _arg_key(v)) for k, v in arg.items()),
                            key=lambda kv: repr(kv[0])))
    return type(arg)


def _clone_args(arg):
    if isinstance(arg, torch.Tensor):
        return arg.detach().clone()
    if isinstance(arg, tuple):
        return tuple(_clone_args(a) for a in arg)
    if isinstance(arg, list):
        return [_clone_args(a) for a in arg]
    if isinstance(arg, dict):
        return {k: _clone_args(v) for k, v in arg.items()}
    if arg is None or isinstance(arg, (str, int, float, bytes, bool)):
        return arg
    raise ValueError(f"Unknown argument type {arg}")


def _copy_into(dst, src):
    if isinstance(src, torch.Tensor):
        dst.copy_(src)
    elif isinstance(src, (tuple, list)):
        for d, s in zip(dst, src):
            _copy_into(d, s)
    elif isinstance(src, dict):
        for k, s in src.items():
            _copy_into(dst[k], s)


def hip_graph_opt(unet, warmup: int = 3):
    """Replace unet.forward by capture-once / copy-inputs / replay, keyed by argument shapes and
    dtypes.  The operators are capture-safe: asynchronous launches on the current stream, scalars
    read on the device, all memory from torch's allocator."""
    lock = threading.Lock()
    cache = {}
    wrapped = unet.forward

    @functools.wraps(wrapped)
    def forward_with_graph(*args, **kwargs):
        key = (_arg_key(args), _arg_key(kwargs))
        if key not in cache:
            with lock:
                if key not in cache:
                    s_args, s_kwargs = _clone_args((args, kwargs))
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.no_grad(), torch.cuda.stream(side):
                        for _ in range(warmup):
                            wrapped(*s_args, **s_kwargs)
                    torch.cuda.current_stream().wait_stream(side)
                    graph = torch.cuda.CUDAGraph()
                    with torch.no_grad(), torch.cuda.graph(graph):
                        s_out = wrapped(*s_args, **s_kwargs)
                    cache[key] = ((s_args, s_kwargs), graph, s_out)
        (s_args, s_kwargs), graph, s_out = cache[key]
        _copy_into((s_args, s_kwargs), (args, kwargs))
        graph.replay()
        return s_out

    forward_with_graph.__self__ = unet
    forward_with_graph._cached = cache
    unet.forward = forward_with_graph
    return unet


cuda_graph_opt = hip_graph_opt   # the reference's name


def example_inputs(batch_size: int, sample_size: int, device, in_channels: int = 4, seed=None):
    """Synthetic UNet inputs of the reference's harness (quantize_sdxl.py:350-373):
    sample rand(B,4,L,L), encoder_hidden_states rand(B,77,2048), timestep 999.,
    text_embeds rand(B,1280), time_ids [[S,S,0,0,S,S]] * B with S = 8 * L, all fp16."""
    g = None
    if seed is not None:
        g = torch.Generator(device="cpu").manual_seed(seed)

    def rand(*shape):
        return torch.rand(*shape, generator=g).to(device=device, dtype=torch.float16)

    px = float(8 * sample_size)
    return dict(
        sample=rand(batch_size, in_channels, sample_size, sample_size),
        timestep=torch.tensor(999., device=device),
        encoder_hidden_states=rand(batch_size, 77, 2048),
        added_cond_kwargs=dict(
            time_ids=torch.tensor([[px, px, 0., 0., px, px]], dtype=torch.float16,
                                  device=device).repeat(batch_size, 1),
            text_embeds=rand(batch_size, 1280)),
    )
