"""The module-swap surface and run harness: counterpart of kernels/quantize_sdxl.py.

    register_qconfig_from_input_files(unet, args, bos, bos_dict)    quantize_sdxl.py:39-139
    convert_to_quantized(unet, ckpt)                                 quantize_sdxl.py:142-150
    quantize_unet(unet, args, ckpt, bos, bos_dict)                   quantize_sdxl.py:154-156
    hip_graph_opt(unet, args=None)  (reference name cuda_graph_opt kept)  quantize_sdxl.py:184-286
    example_inputs(...)                                              quantize_sdxl.py:350-373
    run(unet, args)  static / dynamic / peak memory + timing harness  quantize_sdxl.py:331-484
    layers_roctx_annotate(unet)  profiler ranges per block            quantize_sdxl.py:14-29,387-429

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


def quantize_unet(unet, args, ckpt, bos, bos_dict, w4_kernel=False, swap_glue=False, swap_attention=None,
                  swap_operands=True):
    """The reference's call (quantize_sdxl.py:154-156) plus three options it does not have:
    `w4_kernel=True`: 4-/2-bit weight layers run the packed-W4 INT8 kernels instead of falling back to FP16
    (mixdq_amd.nn.QuantizedLinear.w4_kernel);
    `swap_glue=True`: the stock glue modules BETWEEN the quantized layers -- nn.GroupNorm (+ the nn.SiLU behind
    it), nn.LayerNorm, GEGLU -- are swapped by type for this repo's FP16-output kernels, and (`swap_attention`,
    default: as `swap_glue`) the FP16 attention core for mixdq_attention_f16: same graph, same names, same
    state_dict, one launch per module (mixdq_amd/nn/glue.py; `unswap_glue_modules` undoes it).  `swap_operands`
    (with swap_glue): a swapped producer's launch also writes the INT8 operand of the quantized layers its parent
    hands its output to, and those layers skip their quantize launch -- the same bits, ~600 launches fewer."""
    register_qconfig_from_input_files(unet, args, bos=bos, bos_dict=bos_dict)
    if w4_kernel:
        for mod in unet.modules():
            if getattr(mod, "w_bit", 8) in (2, 4):
                mod.w4_kernel = True
    convert_to_quantized(unet, ckpt)
    if swap_glue or swap_attention:
        from mixdq_amd.nn.glue import swap_glue_modules
        if swap_glue:
            swap_glue_modules(unet, attention=swap_glue if swap_attention is None else bool(swap_attention),
                              operands=bool(swap_operands))
        else:
            raise ValueError("swap_attention=True needs swap_glue=True")


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


def hip_graph_opt(unet, args=None, warmup: int = 3):
    """Replace unet.forward by capture-once / copy-inputs / replay, keyed by argument shapes and
    dtypes.  The operators are capture-safe: asynchronous launches on the current stream, scalars
    read on the device, all memory from torch's allocator.

    Signature of the reference's `cuda_graph_opt(unet, args)` (quantize_sdxl.py:184): `args` (the
    script's argparse namespace) is accepted and, as in the reference, not used."""
    if isinstance(args, int) and not isinstance(args, bool):    # hip_graph_opt(unet, 5): warm-up count
        args, warmup = None, args
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


# ---------------------------------------------------------------------------------------------
# run harness: memory report and profiler ranges (quantize_sdxl.py:331-484)
# ---------------------------------------------------------------------------------------------
def make_memory_friendly(n_bytes: int) -> str:
    return f"{n_bytes / 2 ** 20:.1f} MB"


class MemoryMeter:
    """The reference's three numbers (quantize_sdxl.py:337-338,453-456; kernels/README.md:82-91):
    static = torch.cuda.memory_allocated() once the network is resident, peak =
    max_memory_allocated() after the (graph-captured) runs, dynamic = peak - static."""

    def __init__(self, device=None):
        self.device = device
        torch.cuda.synchronize(device)
        torch.cuda.reset_peak_memory_stats(device)
        self.static = torch.cuda.memory_allocated(device)

    def report(self) -> dict:
        torch.cuda.synchronize(self.device)
        peak = torch.cuda.max_memory_allocated(self.device)
        mb = 2 ** 20
        return dict(static_mb=self.static / mb, dynamic_mb=(peak - self.static) / mb,
                    peak_mb=peak / mb)


def _range_wrap(forward, name):
    @functools.wraps(forward)
    def wrapper(*args, **kwargs):
        torch.cuda.nvtx.range_push(name)        # roctx on ROCm: visible to rocprofv3 --marker-trace
        try:
            return forward(*args, **kwargs)
        finally:
            torch.cuda.nvtx.range_pop()
    wrapper._mixdq_range = name
    return wrapper


def layers_roctx_annotate(unet, every_layer: bool = False):
    """Profiler ranges around the UNet's blocks and a few representative layers -- the reference's
    `layers_nvtx_annotate` (quantize_sdxl.py:387-429) with its names; `torch.cuda.nvtx` emits roctx
    ranges on ROCm.  Ranges are host-side: annotate for EAGER profiling runs, not under a captured
    graph.  Returns the list of range names."""
    mods = {}

    def add(name, getter):
        try:
            mods[name] = getter()
        except (AttributeError, IndexError):
            pass

    add("conv_320_320", lambda: unet.down_blocks[0].resnets[0].conv1)
    add("conv_1280_1280", lambda: unet.down_blocks[2].resnets[0].conv2)
    add("conv_2560_1280", lambda: unet.up_blocks[0].resnets[0].conv1)
    add("linear_640_640", lambda: unet.down_blocks[1].attentions[0].transformer_blocks[0].attn1.to_q)
    add("linear_1280_1280",
        lambda: unet.down_blocks[2].attentions[0].transformer_blocks[0].attn1.to_q)
    add("linear_2048_1280",
        lambda: unet.down_blocks[2].attentions[0].transformer_blocks[0].attn2.to_k)
    for i, blk in enumerate(unet.down_blocks):
        for j, r in enumerate(blk.resnets):
            mods[f"down_block_{i}_resnet_{j}"] = r
        for j, t in enumerate(getattr(blk, "attentions", None) or []):
            mods[f"down_block_{i}_transformers_{j}"] = t
    for i, blk in enumerate(unet.up_blocks):
        for j, r in enumerate(blk.resnets):
            mods[f"up_block_{i}_resnet_{j}"] = r
        for j, t in enumerate(getattr(blk, "attentions", None) or []):
            mods[f"up_block_{i}_transformers_{j}"] = t
    for j, r in enumerate(unet.mid_block.resnets):
        mods[f"mid_block_resnet_{j}"] = r
    for j, t in enumerate(unet.mid_block.attentions[0].transformer_blocks):
        mods[f"mid_block_transformers_{j}"] = t
    if every_layer:
        for name, m in unet.named_modules():
            if name and not any(True for _ in m.children()):
                mods.setdefault(name, m)
    for name, m in mods.items():
        if not hasattr(m.forward, "_mixdq_range"):
            m.forward = _range_wrap(m.forward, name)
    return list(mods)


def run(unet, args=None, batch_size: int = 1, sample_size: int = 128, cuda_graph_only: bool = True,
        profile: bool = False, device="cuda", out=print):
    """The UNet-only leg of the reference's `run(pipeline, args)` (quantize_sdxl.py:331-484): move
    the network to the GPU, report static memory, two eager runs, optional graph capture and two
    replays, dynamic and peak memory, optional 3 range-annotated iterations for the profiler.
    `args` may carry batch_size / cuda_graph_only / profile (the reference's flag names)."""
    batch_size = getattr(args, "batch_size", batch_size)
    cuda_graph_only = getattr(args, "cuda_graph_only", cuda_graph_only)
    profile = getattr(args, "profile", profile)
    unet.to(device)
    meter = MemoryMeter(device)
    out("Static (weights) memory usage: " + make_memory_friendly(meter.static))
    inputs = example_inputs(batch_size, sample_size, device)

    def run_once():
        with torch.no_grad():
            return unet(**inputs)[0]

    latents = run_once()
    latents = run_once()
    if cuda_graph_only:
        hip_graph_opt(unet, args)
        latents = run_once()
        latents = run_once()
    rep = meter.report()
    out("Dynamic (acts) memory usage: " + make_memory_friendly(int(rep["dynamic_mb"] * 2 ** 20)))
    out("Peak (total) memory usage: " + make_memory_friendly(int(rep["peak_mb"] * 2 ** 20)))
    if profile:
        graphed = getattr(unet.forward, "__wrapped__", None)
        if graphed is not None:
            unet.forward = graphed                    # ranges are host-side: profile eagerly
        layers_roctx_annotate(unet)
        for it in range(3):
            torch.cuda.nvtx.range_push(f"iter_{it}")
            run_once()
            torch.cuda.nvtx.range_pop()
        torch.cuda.synchronize()
    return latents, rep


layers_nvtx_annotate = layers_roctx_annotate   # the reference's name (quantize_sdxl.py:387)
