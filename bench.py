#!/usr/bin/env python3
"""Benchmark of the hot path: one SDXL-Turbo UNet forward ("step") with every yaml-listed Linear /
Conv2d running  quantize (HIP) -> INT8 GEMM / implicit-GEMM conv + epilogue (HIP), synthetic
inputs resident in HBM, captured in a hipGraph (the reference measures with CUDA graphs too:
kernels/README.md:94, quantize_sdxl.py:184-286).

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  Under a launcher (torch.distributed.run: RANK / WORLD_SIZE set) this
process is one rank; without one, bench.py starts the N ranks itself as child processes (before any
GPU call in the parent) and relays rank 0's JSON line.  WORLD_SIZE != --gpus is an error.

N = 1 workload: BASELINE.json configs[1] -- W8A8 SDXL-Turbo UNet, 1024x1024 (latent 128), batch 1,
1 step, one MI355X.  N > 1 (default): weak scaling, the same per-GPU batch on every rank
(batch-sharded replicas, rank 0's quantized weights broadcast once over RCCL, no collective in the
step loop).  `--baseline-config 3` is BASELINE.json configs[3] as named: GLOBAL batch 64 sharded
over the N ranks (64 / N per GPU), 4 UNet forwards per image -- strong scaling.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel (an igemm_kernel<BM,BN,BK,CONV> instantiation): the launches of ONE
                forward of the graph that was timed are recorded (mixdq_amd._C.RECORD), replayed
                per kernel instantiation from a hipGraph and bracketed by HIP events on the launch
                stream; algorithmic int8 ops / duration vs the dense INT8 MFMA peak
                (MI355X_MICROARCH.md);
  cpu_baseline  the reference's CPU-runnable path (qdiff fake-quant, Path A) restated in
                oracle/fakequant.py, timed on this box's host cores on a bounded sample;
  fp16          the same UNet graph with nn.Linear / nn.Conv2d in FP16 on the same GPU;
  memory        static / dynamic / peak MB of the W8A8 and the FP16 network, measured as the
                reference's run() does (quantize_sdxl.py:337-338,453-456).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

# dense INT8 MFMA: 2x the BF16 rate per clock (MI355X_MICROARCH.md, Matrix cores table: "I8 ... 2x BF16
# per clock") x its ~2.5 PF dense BF16 figure (Chip-level parameters)
INT8_MFMA_PEAK_TOPS = 5000.0
F16_MFMA_PEAK_TFLOPS = 2500.0     # dense FP16 / BF16 MFMA (same table): the attention core's roof
HBM_PEAK_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1, help="UNet batch per GPU")
    ap.add_argument("--px", type=int, default=1024, choices=[512, 1024])
    ap.add_argument("--w-config", default="weight/uniform_8")
    ap.add_argument("--a-config", default="act/act_8.00")
    ap.add_argument("--no-bos", action="store_true")
    ap.add_argument("--w4-kernel", action="store_true",
                    help="run 4-/2-bit weight layers on the packed-W4 INT8 kernels (BASELINE config "
                         "3, e.g. --w-config weight/weight_4.00 --a-config act/act_7.77); without "
                         "it they fall back to FP16 as in the reference")
    ap.add_argument("--no-graph", action="store_true", help="eager launches (host-bound)")
    ap.add_argument("--no-fuse", action="store_true",
                    help="run the unfused drop-in graph (torch GroupNorm/LayerNorm/GELU + quantize)")
    ap.add_argument("--swap-glue", action="store_true",
                    help="with --no-fuse: quantize_unet(..., swap_glue=True) -- the stock GroupNorm (+ SiLU) / LayerNorm / "
                         "GEGLU modules and the attention core swapped by type for this repo's FP16-output kernels")
    ap.add_argument("--no-fp16", action="store_true", help="skip the FP16 comparison")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-batch8", action="store_true", help="skip the batch-8 object of the default line")
    ap.add_argument("--no-lnchain", action="store_true",
                    help="skip the extra leg that times the fused graph with LayerNorm in the producer GEMM's launch")
    ap.add_argument("--no-dropin", action="store_true",
                    help="skip the module-swap-only (unfused) leg of the default line")
    ap.add_argument("--cpu-seconds", type=float, default=60.0,
                    help="budget of the cpu_baseline leg (layer time): the whole 794-layer inventory on the "
                         "GPU box's host (7-26 s), a MAC-extrapolated sample on a slow one")
    ap.add_argument("--sweep-reps", type=int, default=5)
    ap.add_argument("--tiny", action="store_true",
                    help="small UNet config (tests of the harness itself; not a benchmark)")
    ap.add_argument("--baseline-config", type=int, default=None, choices=[1, 2, 3, 4],
                    help="shorthand for BASELINE.json configs[i]: 1 = W8A8 1024 px batch 1 (the "
                         "default), 2 = W4A8 mixed batch 1 on the W4 kernels, 3 = GLOBAL batch 64 "
                         "sharded over --gpus N (64 / N per GPU), 4 UNet forwards per image: strong "
                         "scaling, 4 = SDXL-base shape of work: batch 8 with classifier-free "
                         "guidance = UNet batch 16 (same UNet shapes; a 'step' stays one UNet "
                         "forward)")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="strong scaling: this many images in total, split evenly over the ranks")
    ap.add_argument("--forwards-per-image", type=int, default=1,
                    help="UNet forwards one image needs (sampling steps); images/s = UNet-batch "
                         "throughput / this")
    ap.add_argument("--vary-timestep", action="store_true",
                    help="a sampling loop as the pipeline runs it: every step replays the ONE captured "
                         "graph with a different timestep VALUE (quantize_sdxl.py:184-286 keys its "
                         "graph cache on shapes / dtypes only)")
    ap.add_argument("--host-only", action="store_true",
                    help="harness check without a GPU: rendezvous, sharding, module swap on the CPU "
                         "and the weight broadcast; no forward, no timing (value = null)")
    ap.add_argument("--unet-rows-per-image", type=int, default=1,
                    help="UNet batch rows one image occupies (2 with classifier-free guidance)")
    ap.add_argument("--profile-ranges", action="store_true",
                    help="after the timed region: 3 eager iterations with roctx ranges per block "
                         "(quantize_sdxl.py:387-429), for rocprofv3 --marker-trace")
    args = ap.parse_args()
    if args.baseline_config == 2:
        args.w_config, args.a_config, args.w4_kernel = "weight/weight_4.00", "act/act_7.77", True
    elif args.baseline_config == 3:
        args.global_batch, args.forwards_per_image = 64, 4
    elif args.baseline_config == 4:
        args.batch, args.unet_rows_per_image = 16, 2     # 8 images x classifier-free guidance
        if args.forwards_per_image > 1:      # the named config: 20 sampling steps per image
            args.vary_timestep = True
    return args


class Cfg:
    def __init__(self, w, a):
        self.w_config, self.a_config = w, a


def time_steps(fn, steps, warmup, device):
    """W untimed steps, then exactly K steps between barrier + synchronize; max over ranks."""
    from mixdq_amd import shard
    with torch.no_grad():
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize(device)
        shard.barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize(device)
        shard.barrier()
        dt = time.perf_counter() - t0
    return shard.max_over_ranks(dt, device)


def layer_shapes(unet, inputs):
    """Input shape of every quantizable layer for these inputs (forward pre-hooks)."""
    from mixdq_amd.unet import quantizable_layers
    shapes, hooks = {}, []
    for name, mod in quantizable_layers(unet).items():
        hooks.append(mod.register_forward_pre_hook(
            lambda m, a, name=name: shapes.__setitem__(name, tuple(a[0].shape))))
    with torch.no_grad():
        unet(**inputs)
    for h in hooks:
        h.remove()
    return shapes


def layer_work(mod, in_shape):
    """(M, N, K) of the GEMM a layer maps to."""
    if isinstance(mod, nn.Linear) or hasattr(mod, "in_features"):
        K, N = mod.in_features, mod.out_features
        M = 1
        for d in in_shape[:-1]:
            M *= d
        return M, N, K
    B, C, H, W = in_shape
    R, S = mod.kernel_size
    P = (H + 2 * mod.padding[0] - R) // mod.stride[0] + 1
    Q = (W + 2 * mod.padding[1] - S) // mod.stride[1] + 1
    return B * P * Q, mod.out_channels, R * S * C


def roofline_sweep(run_eager, device, reps):
    """Per-kernel timing of the INT8 GEMM / conv launches of the graph that was timed: ONE eager
    forward runs with the launch recorder on (every igemm entry-point call with its real device
    tensors: fused q|k|v, k|v, GEMM+GEGLU, residual epilogues, W4), then the recorded launches are
    replayed per kernel instantiation.  Returns per-kernel totals."""
    import mixdq_amd._C as C
    with torch.no_grad():
        run_eager()                                   # builds cached tables / packs, unrecorded
        torch.cuda.synchronize(device)
        C.RECORD = []
        try:
            run_eager()
        finally:
            rec, C.RECORD = C.RECORD, None
        torch.cuda.synchronize(device)
    groups = {}
    for kind, (M, N, K, k_align), w4, replay in rec:
        if kind == "attention":          # FP16 attention core: M = B * heads * Tq, N = Tkv, K = head_dim
            kern = "attn_short_kernel" if N <= 128 else "attn_fwd_kernel"     # csrc/attention.hip's rule
            g_ = groups.setdefault(f"{kern}<Tkv={N}>", dict(fns=[], ops=0.0, bytes=0.0, f16=True))
            g_["fns"].append(replay)
            g_["ops"] += 4.0 * M * N * K                       # Q K^T and P V
            g_["bytes"] += 2.0 * (2 * M * K) + 0.0             # q read + o written (k / v are re-read per q block)
            continue
        if kind == "linear_grouped":     # N = the members' total; mixdq_qlinear_w8a8_grouped's rule
            cid = 37 if M <= 64 else 35
        elif kind == "linear_attn":      # to_q + cross-attention: always the 64x128 8-wave tile
            cid = 41
        elif kind == "linear_ln":        # GEMM + residual + LayerNorm + quantize: csrc/igemm_ln.hip's rule
            cid = int(C._lib.mixdq_qlinear_ln_select_id(M, N, K))
        elif kind == "linear_f16in":     # quantize-in-prologue GEMM: csrc/igemm_aq.hip's rule
            cid = int(C._lib.mixdq_qlinear_f16in_select_id(M, N, K, int(w4)))
        else:
            cid = C.igemm_select_id(M, N, k_align, K, w4=w4, geglu=kind == "linear_geglu")
        bm, bn, bk, st = C.IGEMM_CONFIGS.get(cid, (0, 0, 0, 0))
        kname = f"igemm_kernel<{bm},{bn},{bk},{st},{kind}{',w4' if w4 else ''}>#cfg{cid}"
        if kind.startswith("conv_halo"):        # 3x3 conv with the input halo resident in LDS
            th, tw, bn = C.HALO_TILES[int(kind[9:])]
            kname = f"conv3x3_halo_kernel<{th},{tw},{bn}>"
        g_ = groups.setdefault(kname, dict(fns=[], ops=0.0, bytes=0.0))
        g_["fns"].append(replay)
        g_["ops"] += 2.0 * M * N * K
        out_bytes = M * (N // 2) if kind == "linear_geglu" else 2 * M * N
        g_["bytes"] += M * K + N * K // (2 if w4 else 1) + out_bytes   # algorithmic: A + W + D
    # Each group is captured (model order) in a hipGraph so no host gap sits between launches, and
    # `reps` replays are bracketed with HIP events recorded on the launch stream (torch's current
    # stream is the stream the kernels are launched on).
    stats = {}
    with torch.no_grad():
        for kname, g_ in groups.items():
            for fn in g_["fns"]:
                fn()
            torch.cuda.synchronize(device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for fn in g_["fns"]:
                    fn()
            graph.replay()
            torch.cuda.synchronize(device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                graph.replay()
            e1.record()
            torch.cuda.synchronize(device)
            stats[kname] = dict(ms=e0.elapsed_time(e1), ops=g_["ops"] * reps,
                                bytes=g_["bytes"] * reps, launches=len(g_["fns"]) * reps,
                                f16=bool(g_.get("f16")))
            del graph
    return stats


def roofline_object(roof_stats, B, reps, px, tiny):
    """The `roofline` object of the JSON line from roofline_sweep's per-kernel totals.  The dominant
    kernel is chosen among the INT8 GEMM / conv instantiations (the metric's roof is the INT8 MFMA
    peak); the FP16 attention core is listed beside them under `per_kernel` against ITS roof."""
    i8 = {k: v for k, v in roof_stats.items() if not v.get("f16")}
    dom = max(i8, key=lambda k: i8[k]["ms"])
    s = i8[dom]
    achieved = s["ops"] / (s["ms"] * 1e-3) / 1e12
    tot_ops = sum(v["ops"] for v in i8.values())
    tot_ms = sum(v["ms"] for v in i8.values())
    # HBM-side bytes per launch and MFMA-busy fraction of THIS instantiation, from the rocprofv3
    # --pmc passes (tools/pmc_r04.sh: separate FETCH_SIZE / WRITE_SIZE / SQ runs; FETCH_SIZE x2 on
    # gfx950) -- keyed by the kernel's own name, null when it was not collected
    traffic = mfma_util = None
    pmc_all = {}
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    # The counter columns are NOT measured by this run: they are the committed results of the builder's
    # rocprofv3 --pmc passes over these kernels (a counter pass serialises the launches and cannot run inside
    # the timed command).  The line says so (`traffic_source`) and drops them when the table was collected on
    # other kernel sources than the library that is running (`csrc_sha16` of the table != the tree's).
    traffic_source = {"file": "profiles/pmc_traffic.json", "measured_by_this_run": False}
    if os.path.exists(pmc) and px == 1024 and not tiny:
        with open(pmc) as f:      # sections by batch size of the probed launches (bs1, bs8)
            table = json.load(f)
        prov = table.get("_provenance", {})
        traffic_source.update(prov)
        traffic_source["csrc_sha16_running"] = csrc_sha16()
        traffic_source["stale"] = (traffic_source["csrc_sha16_running"] is None
                                   or prov.get("csrc_sha16") != traffic_source["csrc_sha16_running"])
        if not traffic_source["stale"]:
            pmc_all = table.get(f"bs{B}", {})
        entry = pmc_all.get(dom) or pmc_all.get(dom.split("#")[0], {})     # "...#cfgNN" where a tile has several wave layouts
        traffic, mfma_util = entry.get("hbm_bytes_per_launch"), entry.get("mfma_util")
    per_kernel = {}
    for k, v in sorted(roof_stats.items()):
        peak = F16_MFMA_PEAK_TFLOPS if v.get("f16") else INT8_MFMA_PEAK_TOPS
        tops = v["ops"] / (v["ms"] * 1e-3) / 1e12
        per_kernel[k] = {"ms_per_step": v["ms"] / reps, "launches": v["launches"] // reps,
                         "avg_launch_us": 1e3 * v["ms"] / v["launches"], "tops": tops,
                         "frac": tops / peak, "peak": peak}
        e = pmc_all.get(k) or (None if any(n.startswith(k.split("#")[0] + "#") for n in pmc_all)
                               else pmc_all.get(k.split("#")[0]))
        if e:
            per_kernel[k].update({kk: e[kk] for kk in ("mfma_util", "hbm_bytes_per_launch", "valu_util")
                                  if kk in e})
    return {
        "bound": "mfma", "kernel": dom, "achieved": achieved, "peak": INT8_MFMA_PEAK_TOPS,
        "unit": "TFLOP/s", "frac": achieved / INT8_MFMA_PEAK_TOPS, "traffic": traffic,
        "mfma_util": mfma_util, "traffic_source": traffic_source,
        "clock": "cold replay of the recorded launches, each on its own weights, without the prefetch payload "
                 "that precedes it in the step (HIP events); in_step_* = the same kernel inside an eager "
                 "forward of the benchmarked network, prefetch payloads included (torch.profiler durations)",
        "launches_per_step": s["launches"] // reps,
        "avg_launch_us": 1e3 * s["ms"] / s["launches"],
        "ops_per_launch": s["ops"] / s["launches"],
        "algorithmic_bytes_per_launch": s["bytes"] / s["launches"],
        "hbm_frac_of_8TBs": s["bytes"] / (s["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "all_igemm": {"tops": tot_ops / (tot_ms * 1e-3) / 1e12,
                      "ms_per_step": tot_ms / reps,
                      "launches_per_step": sum(v["launches"] for v in i8.values()) // reps,
                      "int8_ops_per_step": tot_ops / reps},
        "per_kernel": per_kernel,
    }


def under_rocprof() -> bool:
    """Is this process running under rocprofv3?  (Then torch.profiler -- a second tracer in the process -- is not
    started: the legs that count kernels with it are skipped, whatever the flags say.  ADVICE r4.)"""
    return bool(os.environ.get("ROCP_TOOL_LIBRARIES") or os.environ.get("ROCPROFILER_REGISTER_FORCE_LOAD")
                or "rocprofiler" in os.environ.get("LD_PRELOAD", ""))


def csrc_sha16():
    """sha256 (first 16 hex digits) of the kernel sources the LOADED library was built from: the hash
    mixdq_amd/build.py embedded in it (mixdq_build_csrc_sha16), read back through the library -- not a hash of the
    tree on disk, which says nothing about a stale .so or one named by MIXDQ_HIP_LIB.  None: a library that does
    not carry one (a build of an older tree); provenance is then unknown and the counter columns are dropped."""
    import ctypes
    import mixdq_amd._C as C
    try:
        fn = C._lib.mixdq_build_csrc_sha16
    except AttributeError:
        return None
    fn.restype = ctypes.c_char_p
    v = fn()
    return v.decode() if v else None


def rocprof_kernel_avg(kname, B):
    """The rocprofv3 --kernel-trace --stats average of the dominant kernel, from the COMMITTED summary of the same
    command (profiles/r06_bench_kernel_stats_bs{B}.csv; a trace cannot run inside the timed command), beside the
    live HIP-event clock: {rocprof_avg_launch_us, rocprof_calls, frac_rocprof, rocprof_source}, or None."""
    import csv
    path = os.path.join(ROOT, "profiles", f"r06_bench_kernel_stats_bs{B}.csv")
    if not os.path.exists(path):
        return None
    pat = kernel_name_pattern(kname)
    best = None
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Name") or row.get("KernelName") or ""
            if pat in name:
                calls = int(float(row.get("Calls", 0) or 0))
                tot = float(row.get("TotalDurationNs", 0) or 0)
                if calls and (best is None or tot > best[1]):
                    best = (calls, tot)
    if not best:
        return None
    return {"rocprof_avg_launch_us": best[1] / best[0] / 1e3, "rocprof_calls": best[0],
            "rocprof_source": os.path.relpath(path, ROOT) + " (committed summary of the same command; not measured by this run)"}


def kernel_name_pattern(kname):
    """bench.py kernel name -> substring of the (demangled) device kernel name torch.profiler reports."""
    import re
    import mixdq_amd._C as C
    m = re.match(r"igemm_kernel<(\d+),(\d+),(\d+),(\d+),(\w+?)(,w4)?>#cfg(\d+)", kname)
    if m and int(m.group(7)) == 71:      # the persistent four-phase kernel (csrc/igemm_pp.h)
        return "igemm_pp_kernel<true>" if "geglu" in m.group(5) else "igemm_pp_kernel<false>"
    if m:
        wm, wn, ks, mt = C.IGEMM_WAVES[int(m.group(7))]
        return "igemm_kernel<%s, %s, %s, %s, %d, %d, " % (m.group(1), m.group(2), m.group(3), m.group(4), wm, wn)
    m = re.match(r"conv3x3_halo_kernel<(\d+),(\d+),(\d+)>", kname)
    if m:
        return "conv3x3_halo_kernel<%s, %s, %s, " % m.groups()
    return kname.split("<")[0]


def in_step_kernel_times(fn, device):
    """One torch.profiler pass over fn() (an eager forward of the benchmarked graph's launches): {device
    kernel name: (launches, total us)}, or None.  Traced durations over-state kernels of a few us by 1.5-3 us and long
    ones by ~0.3; they are the kernel INSIDE the step: warm scalar caches, weights where the
    prefetch left them."""
    try:
        from torch.profiler import ProfilerActivity, profile
        with torch.no_grad():
            fn()
            torch.cuda.synchronize(device)
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                fn()
                torch.cuda.synchronize(device)
        out = {}
        for e in prof.events():
            if not str(getattr(e, "device_type", "")).endswith("CUDA") or "Memcpy" in e.name or "Memset" in e.name:
                continue
            n, t = out.get(e.name, (0, 0.0))
            out[e.name] = (n + 1, t + float(getattr(e, "device_time", 0.0) or getattr(e, "cuda_time", 0.0)))
        return out or None
    except Exception:
        return None


def count_kernels(fn, device):
    """GPU kernels one call of fn() launches (torch.profiler / roctracer), or None if unavailable."""
    try:
        from torch.profiler import ProfilerActivity, profile
        with torch.no_grad():
            fn()
            torch.cuda.synchronize(device)
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                fn()
                torch.cuda.synchronize(device)
        n = sum(1 for e in prof.events()
                if str(getattr(e, "device_type", "")).endswith("CUDA") and "Memcpy" not in e.name
                and "Memset" not in e.name)
        return n or None
    except Exception:
        return None


def cpu_fake_quant_baseline(seconds_budget):
    """Path A (qdiff fake-quant) of the 794-layer inventory at 512 px, batch 1, FP32 on the host
    cores: the WHOLE inventory where the host is fast enough (the GPU box: 128 threads, 7-26 s), else a
    sample spread uniformly over the model within `seconds_budget` with the rest extrapolated by
    multiply-accumulates -- the returned object says which (whole_inventory, measured_mac_frac)."""
    from oracle.fakequant import quant_layer_forward
    from mixdq_amd.calib import ActRange, weight_delta
    from mixdq_amd.quantize_sdxl import example_inputs
    from mixdq_amd.unet import SDXLUNet, quantizable_layers
    with torch.device("meta"):
        unet = SDXLUNet().half()
        inp = example_inputs(1, 64, "meta")
    shapes = layer_shapes(unet, inp)
    layers = list(quantizable_layers(unet).items())
    stride = 8
    t_total, n_done, macs_done, macs_all = 0.0, 0, 0.0, 0.0
    for name, mod in layers:
        M, N, K = layer_work(mod, shapes[name])
        macs_all += float(M) * N * K
    torch.manual_seed(0)
    torch.randn(512, 512) @ torch.randn(512, 512)          # spin the thread pool up, untimed
    t_begin = time.perf_counter()
    # every 8th layer first, then the other residues mod 8 while the budget lasts: a fast host
    # (the GPU box: 128 threads) ends up timing the whole inventory, a slow one a uniform sample
    order = [i for off in range(stride) for i in range(off, len(layers), stride)]
    for idx in order:
        name, mod = layers[idx]
        shp = shapes[name]
        x = torch.randn(shp)
        w = torch.randn(tuple(mod.weight.shape)) * 0.02
        b = torch.zeros(w.shape[0]) if mod.bias is not None else None
        rng = ActRange()
        rng.update(x)
        a_delta, a_zp = rng.params(2)
        w_delta = weight_delta(w, 8)
        kw = None
        if isinstance(mod, nn.Conv2d):
            kw = dict(stride=mod.stride, padding=mod.padding, dilation=mod.dilation, groups=1)
        with torch.no_grad():
            t0 = time.perf_counter()
            quant_layer_forward(x, w, b, w_delta, a_delta, a_zp, 8, 8, kw)
            t_total += time.perf_counter() - t0
        n_done += 1
        M, N, K = layer_work(mod, shp)
        macs_done += float(M) * N * K
        if t_total > seconds_budget or time.perf_counter() - t_begin > 3 * seconds_budget:
            break
    # The budget (default 60 s of layer time) is sized so that the GPU box's host (128 threads: 7-26 s
    # per forward) runs the WHOLE inventory: measured_mac_frac = 1, nothing extrapolated.  A slower
    # host stops at the budget; what it did not reach is extrapolated by multiply-accumulates (the
    # sample is spread uniformly over the model, but layers differ by 1000x in work), and the line
    # says how much of the number was measured.
    frac = macs_done / macs_all
    est_forward_s = t_total / max(frac, 1e-9)
    whole = n_done == len(layers)
    return dict(value=1.0 / est_forward_s, unit="images/s", cores=torch.get_num_threads(),
                kind="port", seconds_per_forward_est=est_forward_s,
                measured_s=t_total, measured_layers=n_done, measured_mac_frac=frac,
                extrapolated_s=est_forward_s - t_total, whole_inventory=whole,
                sample=f"qdiff fake-quant (Path A) W8A8 512px bs1 FP32 on CPU: {n_done} of "
                       f"{len(layers)} layers ({100 * frac:.1f}% of the MACs) in {t_total:.1f} s"
                       + ("" if whole else ", the rest extrapolated by MACs")
                       + f"; host os.cpu_count()={os.cpu_count()}")


def spawn_ranks(n: int) -> int:
    """`--gpus N` without a launcher: start the N ranks as fresh child processes (one per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) BEFORE this process has made
    any GPU call -- it never does -- relay rank 0's JSON line, and return non-zero if any rank
    failed.  (Never re-exec a process that has touched the GPU.)

    The ranks run in a process group of their own: when one of them dies, or this process is told
    to stop (SIGTERM / SIGINT, e.g. `timeout 600 python bench.py --gpus 2`), the rest are ended too
    instead of sitting in a rendezvous until torch's timeout while holding the GPUs.  Of rank 0's
    stdout only the last line that parses as JSON goes to stdout; everything else goes to stderr."""
    import signal
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs, pgid = [], None

    def end_all(sig=signal.SIGTERM):
        if pgid is not None:
            try:
                os.killpg(pgid, sig)
            except ProcessLookupError:
                pass

    def on_signal(signum, _frame):
        end_all()
        raise SystemExit(128 + signum)

    old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    out0 = b""
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            pr = subprocess.Popen(
                [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                stdout=subprocess.PIPE if r == 0 else sys.stderr,
                preexec_fn=(lambda g=(pgid or 0): os.setpgid(0, g)))   # rank 0 founds the group
            if r == 0:
                pgid = pr.pid
            procs.append(pr)
        import threading
        buf = []
        reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
        reader.start()
        codes = [None] * n
        while any(c is None for c in codes):
            for r, pr in enumerate(procs):
                if codes[r] is None:
                    codes[r] = pr.poll()
            if any(c not in (None, 0) for c in codes):
                end_all()                      # a rank failed: the others would wait for it forever
                for r, pr in enumerate(procs):
                    if codes[r] is None:
                        try:
                            codes[r] = pr.wait(timeout=30)
                        except subprocess.TimeoutExpired:
                            end_all(signal.SIGKILL)
                            codes[r] = pr.wait()
                break
            time.sleep(0.2)
        reader.join(timeout=10)
        out0 = buf[0] if buf else b""
    finally:
        if any(pr.poll() is None for pr in procs):
            end_all()
        for sg, h in old.items():
            signal.signal(sg, h)
    lines = out0.decode(errors="replace").splitlines()
    last_json = None
    for i in range(len(lines) - 1, -1, -1):
        try:
            json.loads(lines[i])
            last_json = i
            break
        except ValueError:
            continue
    for i, ln in enumerate(lines):
        print(ln, file=sys.stdout if i == last_json else sys.stderr)
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def host_only(args, rank, world):
    """Harness check on the CPU (tests/test_bench_cpu.py): the N > 1 plumbing of this file --
    rendezvous, batch sharding, module swap, the one-time weight broadcast -- with no forward and
    no timing.  The JSON line says so: value = null."""
    from mixdq_amd import shard
    from mixdq_amd.calib import calibrate, precompute_bos
    from mixdq_amd.quantize_sdxl import example_inputs, quantize_unet
    from mixdq_amd.unet import SDXLUNet, init_synthetic_weights, quantizable_layers
    assert args.tiny, "--host-only builds the network on the CPU: --tiny only"
    strong = args.global_batch is not None
    if strong:
        lo, hi = shard.shard_range(args.global_batch, rank, world)
        B, global_batch = hi - lo, args.global_batch
    else:
        B, global_batch = args.batch, world * args.batch
    unet = init_synthetic_weights(SDXLUNet(TINY_CFG)).eval()
    inputs = example_inputs(1, 8, "cpu", seed=42 + rank)
    inputs = {k: (v.float() if torch.is_tensor(v) else {a: b.float() for a, b in v.items()})
              for k, v in inputs.items()}
    ckpt = calibrate(unet, [inputs])
    bos_dict = precompute_bos(unet.half(), inputs["encoder_hidden_states"].half())
    names = list(quantizable_layers(unet))
    quantize_unet(unet, Cfg({n: 8 for n in names},
                            {n: 8 for n in names if n not in ("conv_in", "conv_out")}),
                  ckpt, bos=True, bos_dict=bos_dict)
    bcast = shard.broadcast_module_state(unet, src=0)
    shard.barrier()
    if rank == 0:
        fpi = max(1, args.forwards_per_image)
        print(json.dumps({
            "metric": "sdxl_turbo_unet_w8a8_images_per_sec", "value": None, "unit": "images/s",
            "n_gpus": world, "steps": 0, "warmup": 0, "ms_per_step": None,
            "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "int8", "data": "synthetic", "host_only": True,
            "config": {"workload": "harness check (no forward)", "global_batch": global_batch,
                       "per_gpu_batch": B, "unet_forwards_per_image": fpi,
                       "parallelism": f"dp{world} (batch-sharded replicas, no step-loop collective)"},
            "weight_broadcast_bytes": bcast}))


TINY_CFG = dict(block_out_channels=(32, 64, 128), transformer_layers_per_block=(0, 1, 2),
                mid_transformer_layers=1, head_dim=16, cross_attention_dim=2048,
                time_embed_dim=128, addition_time_embed_dim=16,
                projection_class_embeddings_input_dim=1280 + 96, norm_num_groups=8)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))      # this process makes no GPU call
    from mixdq_amd import shard
    if not args.host_only:
        assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU fallback)"
    # MIXDQ_DIST_BACKEND=gloo + MIXDQ_SHARE_DEVICE=1: several ranks on ONE GPU, to exercise the
    # N > 1 code path on a single-GPU box (harness test only).
    share = os.environ.get("MIXDQ_SHARE_DEVICE") == "1"
    rank, local_rank, world = shard.init_distributed(
        os.environ.get("MIXDQ_DIST_BACKEND") or ("gloo" if args.host_only else None))
    if world != args.gpus:      # never report an N-GPU request as a 1-GPU number
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.host_only:
        return host_only(args, rank, world)
    device = torch.device("cuda", 0 if share else local_rank)
    torch.cuda.set_device(device)

    import mixdq_amd._C as C
    from mixdq_amd import cfgs
    from mixdq_amd.calib import calibrate, precompute_bos
    from mixdq_amd.nn import QuantizedConv2d, QuantizedLinear
    from mixdq_amd.quantize_sdxl import (MemoryMeter, example_inputs, hip_graph_opt,
                                         layers_roctx_annotate, quantize_unet)
    from mixdq_amd.unet import build_unet

    L = args.px // 8
    strong = args.global_batch is not None
    if strong:      # fixed total work: this rank's contiguous share of the global batch
        lo, hi = shard.shard_range(args.global_batch, rank, world)
        B, global_batch = hi - lo, args.global_batch
        assert B > 0, f"global batch {args.global_batch} leaves rank {rank} of {world} empty"
    else:           # fixed per-GPU work
        B, global_batch = args.batch, world * args.batch
    fpi = max(1, args.forwards_per_image)
    t_setup = time.perf_counter()
    unet = build_unet(device, cfg=TINY_CFG if args.tiny else None)
    fp16_meter = MemoryMeter(device)              # static = the FP16 network, resident
    inputs = example_inputs(B, L, device, seed=42 + rank)
    # static per-tensor scales come from a calibration set, not from the batch that is served
    # (the reference loads them from a checkpoint): two images of this rank's inputs -- the FP16
    # calibration forward runs on MIOpen / hipBLASLt, whose per-shape kernel search at batch 64
    # took longer than everything else in this file together
    def first_rows(v, n):
        if torch.is_tensor(v):
            return v[:n] if v.dim() > 0 and v.shape[0] == B else v
        return {a: first_rows(b, n) for a, b in v.items()} if isinstance(v, dict) else v
    calib_inputs = {k: first_rows(v, 2) for k, v in inputs.items()} if B > 2 else inputs
    ckpt = calibrate(unet, [calib_inputs], bos=not args.no_bos)
    bos_dict = precompute_bos(unet, inputs["encoder_hidden_states"])

    # --vary-timestep: the pipeline's sampling loop -- each step hands the UNet another timestep
    # VALUE (a 0-dim device tensor, as the scheduler does); the graph cache keys on shapes and
    # dtypes, so all steps replay ONE captured graph after copying the value into its input
    n_ts = max(2, fpi) if args.vary_timestep else 1
    timesteps = [torch.tensor(999.0 - i * (998.0 / max(1, n_ts - 1)), device=device)
                 for i in range(n_ts)]
    step_no = [0]

    def run_once():
        if args.vary_timestep:
            inputs["timestep"] = timesteps[step_no[0] % n_ts]
            step_no[0] += 1
        return unet(**inputs)[0]

    rows_per_image = max(1, args.unet_rows_per_image)
    fp16, memory = None, {}
    if not args.no_fp16:
        # (a) the reference's comparison point: the same graph, stock PyTorch FP16 ops throughout
        if not args.no_graph:
            hip_graph_opt(unet)
        dt = time_steps(run_once, args.steps, args.warmup, device)
        fp16 = dict(ms_per_step=1e3 * dt / args.steps,
                    images_per_s=global_batch * args.steps / dt / fpi / rows_per_image)
        memory["fp16"] = fp16_meter.report()
        if not args.no_graph:
            unet.forward = unet.forward.__wrapped__   # drop the FP16 graph
        # (b) for transparency: FP16 GEMMs / convs by PyTorch, but with this repo's fused
        #     GroupNorm / LayerNorm / GEGLU glue (fp16 outputs), so the INT8-vs-FP16 kernel effect
        #     can be read separately from the glue effect
        if not args.no_fuse:
            unet.set_fused(True)
            if not args.no_graph:
                hip_graph_opt(unet)
            dt = time_steps(run_once, args.steps, args.warmup, device)
            fp16["fused_glue_ms_per_step"] = 1e3 * dt / args.steps
            if not args.no_graph:
                unet.forward = unet.forward.__wrapped__
            unet.set_fused(False)

    if args.tiny:
        from mixdq_amd.unet import quantizable_layers
        names = list(quantizable_layers(unet))
        w_cfg = {n: 8 for n in names}
        a_cfg = {n: 8 for n in names if n not in ("conv_in", "conv_out")}
    else:
        w_cfg, a_cfg = cfgs.load(args.w_config), cfgs.load(args.a_config)
    quantize_unet(unet, Cfg(w_cfg, a_cfg), ckpt, bos=not args.no_bos, bos_dict=bos_dict,
                  w4_kernel=args.w4_kernel, swap_glue=bool(args.swap_glue and args.no_fuse))
    del ckpt
    unet.set_fused(not args.no_fuse)
    bcast_bytes = shard.broadcast_module_state(unet, src=0)
    qmods = [m for m in unet.modules() if isinstance(m, (QuantizedLinear, QuantizedConv2d))]
    n_accel = sum(m.valid_for_acceleration for m in qmods)
    n_w4 = sum(m.valid_for_acceleration and getattr(m, "w_packed4", False) for m in qmods)
    with torch.no_grad():
        run_once()          # one eager forward: packed q|k|v / k|v operands, conv border tables and
    torch.cuda.synchronize(device)   # persistent K/V buffers are static data, built before the meter
    import gc
    gc.collect()
    if hasattr(torch._C, "_cuda_clearCublasWorkspaces"):
        torch._C._cuda_clearCublasWorkspaces()   # hipBLASLt workspaces of the FP16 legs above: the
    torch.cuda.empty_cache()                     # quantized step launches no vendor-library kernel
    weight_bytes = sum(b.numel() * b.element_size() for b in unet.buffers()) + sum(
        p.numel() * p.element_size() for p in unet.parameters())
    q_meter = MemoryMeter(device)                 # static = the quantized network, resident
    shard.barrier()

    eager_forward = unet.forward
    if not args.no_graph:
        hip_graph_opt(unet)
    setup_s = time.perf_counter() - t_setup
    dt = time_steps(run_once, args.steps, args.warmup, device)
    ms = 1e3 * dt / args.steps
    value = global_batch * args.steps / dt / fpi / rows_per_image
    graphs_cached = None if args.no_graph else len(unet.forward._cached)
    memory["w4a8_mixed" if args.w4_kernel else "w8a8"] = q_meter.report()

    roof_stats = None
    in_step = None
    if not args.no_roofline and rank == 0:
        roof_stats = roofline_sweep(lambda: eager_forward(**inputs), device, args.sweep_reps)
        if not under_rocprof():
            # (an EAGER forward of the timed graph's launches -- tracing a hipGraph replay through torch.profiler
            #  crashed the process on ROCm 7.2; and not under rocprofv3: two tracers in one process)
            in_step = in_step_kernel_times(lambda: eager_forward(**inputs), device)

    # ---- batch 8 beside the headline (north_star: "bs=1/8 on 1 GPU"; also the per-GPU shard of
    #      configs[3] on 8 GPUs): the same network and graph machinery on a batch-8 input, timed AFTER
    #      the main region, with its own dominant kernel's roofline.  Default N = 1 run only.
    batch8 = None
    default_line = (world == 1 and B == 1 and not strong and not args.tiny and args.px == 1024
                    and not args.no_graph)
    if default_line and not args.no_batch8:
        inputs8 = example_inputs(8, L, device, seed=1042)
        with torch.no_grad():
            eager_forward(**inputs8)                   # persistent K/V buffers of this shape
        torch.cuda.synchronize(device)
        k8 = max(5, args.steps // 2)
        dt8 = time_steps(lambda: unet(**inputs8)[0], k8, 2, device)
        batch8 = {"ms_per_step": 1e3 * dt8 / k8, "images_per_s": 8 * k8 / dt8, "steps": k8,
                  "workload": f"sdxl_turbo_unet_{'w4a8_mixed' if args.w4_kernel else 'w8a8'}_{args.px}px_bs8_1step"}
        if not args.no_roofline:
            r8 = roofline_object(roofline_sweep(lambda: eager_forward(**inputs8), device,
                                                max(2, args.sweep_reps // 2)),
                                 8, max(2, args.sweep_reps // 2), args.px, args.tiny)
            batch8["roofline"] = {k: r8[k] for k in ("kernel", "achieved", "peak", "frac", "traffic",
                                                     "mfma_util", "launches_per_step", "avg_launch_us",
                                                     "algorithmic_bytes_per_launch", "all_igemm")}
            batch8["roofline"]["per_kernel"] = r8["per_kernel"]
        del inputs8

    # ---- what the module swap ALONE gives (mixdq_extension._C replaced under
    #      an unchanged graph): the same quantized network with this repo's producer fusions off --
    #      one quantize launch per layer, torch GroupNorm / LayerNorm / GELU / SDPA glue -- in a hipGraph.
    dropin = None
    if default_line and not args.no_dropin and not args.no_fuse and not under_rocprof():
        unet.forward = eager_forward
        unet.set_fused(False)
        with torch.no_grad():
            eager_forward(**inputs)
        torch.cuda.synchronize(device)
        n_unfused = count_kernels(lambda: eager_forward(**inputs), device)
        hip_graph_opt(unet)
        kd = max(5, args.steps // 2)
        dtd = time_steps(run_once, kd, 2, device)
        unet.forward = eager_forward
        dropin = {"dropin_unfused_ms_per_step": 1e3 * dtd / kd, "dropin_unfused_kernels_per_step": n_unfused}
        # ... and with quantize_unet(..., swap_glue=True) (mixdq_amd/nn/glue.py): the stock nn.GroupNorm (+ SiLU),
        #     nn.LayerNorm and GEGLU modules swapped by type for this repo's FP16-output kernels, one launch per
        #     module, same graph -- first with the attention core left to PyTorch's SDPA, then with it on
        #     mixdq_attention_f16 as well (what swap_glue=True does by default)
        from mixdq_amd.nn.glue import swap_glue_modules, unswap_glue_modules
        swapped = {}
        #     -- both with every layer still running its own quantize launch (operands=False) -- and last as
        #     swap_glue=True does it by default: the producers' launches also write the INT8 operand of the quantized
        #     layers behind them (nn/glue.py OPERAND_PAIRS), whose quantize launches disappear
        for att, ops, key in ((False, False, "dropin_glue_torch_sdpa"), (True, False, "dropin_glue_own_quantize"),
                              (True, True, "dropin_glue")):
            for k_, v_ in swap_glue_modules(unet, attention=att, operands=ops).items():
                swapped[k_] = swapped.get(k_, 0) + v_
            with torch.no_grad():
                eager_forward(**inputs)
            torch.cuda.synchronize(device)
            n_k = count_kernels(lambda: eager_forward(**inputs), device)
            hip_graph_opt(unet)
            dtg = time_steps(run_once, kd, 2, device)
            unet.forward = eager_forward
            dropin[key + "_ms_per_step"] = 1e3 * dtg / kd
            dropin[key + "_kernels_per_step"] = n_k
        dropin["dropin_glue_swapped_modules"] = swapped
        dropin["dropin_glue_attention"] = ("dropin_unfused / dropin_glue_torch_sdpa: torch F.scaled_dot_product_attention "
                                           "(its AOTriton kernel is also called attn_fwd); dropin_glue_own_quantize, dropin_glue and the headline: "
                                           "this repo's mixdq_attention_f16 (csrc/attention.hip)")
        unswap_glue_modules(unet)
        unet.set_fused(True)
        n_fused = count_kernels(lambda: eager_forward(**inputs), device)
        dropin["kernels_per_step"] = n_fused
    # ---- the same fused graph with every eligible LayerNorm riding in its producer GEMM's launch (DESIGN.md 3.13:
    #      off by default because it is time-neutral): its step time and kernel count beside the headline's
    ln_in_gemm = None
    if default_line and not args.no_lnchain and not args.no_fuse and not under_rocprof():
        import mixdq_amd.unet as U_
        saved_chain = U_.LN_CHAIN
        try:
            U_.LN_CHAIN = not saved_chain
            unet.forward = eager_forward
            with torch.no_grad():
                eager_forward(**inputs)
            torch.cuda.synchronize(device)
            n_alt = count_kernels(lambda: eager_forward(**inputs), device)
            hip_graph_opt(unet)
            kl = max(5, args.steps // 2)
            dtl = time_steps(run_once, kl, 2, device)
            ln_in_gemm = {"enabled_in_headline": bool(saved_chain), "alternative_ms_per_step": 1e3 * dtl / kl,
                          "alternative_kernels_per_step": n_alt,
                          "alternative": "LayerNorm + quantize in the producer GEMM's launch" if not saved_chain
                                         else "every LayerNorm a launch of its own"}
        finally:
            U_.LN_CHAIN = saved_chain
            unet.forward = eager_forward
    if args.profile_ranges and rank == 0:
        unet.forward = eager_forward
        layers_roctx_annotate(unet)
        with torch.no_grad():
            for it in range(3):
                torch.cuda.nvtx.range_push(f"iter_{it}")
                run_once()
                torch.cuda.nvtx.range_pop()
        torch.cuda.synchronize(device)
    shard.barrier()

    if rank != 0:
        return
    kind = "w4a8_mixed" if args.w4_kernel else "w8a8"
    out = {
        "metric": f"sdxl_turbo_unet_{kind}_images_per_sec",
        "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "int8", "data": "synthetic",
        "config": {
            "workload": ("TINY harness network (not the SDXL UNet), " if args.tiny else "") +
                        f"sdxl_turbo_unet_{kind}_{args.px}px_bs{global_batch if strong else B}"
                        f"_{fpi}step" + ("_sharded" if strong else ""),
            "global_batch": global_batch, "per_gpu_batch": B, "px": args.px, "latent": L,
            "unet_forwards_per_image": fpi, "unet_rows_per_image": rows_per_image,
            "vary_timestep": bool(args.vary_timestep), "graphs_cached": graphs_cached,
            "w_config": args.w_config, "a_config": args.a_config, "bos": not args.no_bos,
            "parallelism": f"dp{world} (batch-sharded replicas, no step-loop collective)",
            "hip_graph": not args.no_graph, "producer_fusions": not args.no_fuse,
            "swap_glue": bool(args.swap_glue and args.no_fuse),
            "accelerated_layers": n_accel, "w4_kernel_layers": n_w4,
            "quantizable_layers": len(qmods),
            "epilogue_variant": "B" if C.FLAGS & 1 else "A",
        },
        "unet_step_latency_ms": ms,
        "weights_mb": weight_bytes / 2 ** 20,
        "memory": memory,
        "weight_broadcast_bytes": bcast_bytes,
        "setup_s": setup_s,
    }
    if fp16:
        out["fp16"] = fp16
        # three ratios against the FP16 network on this GPU; the like-for-like one is the headline:
        #   speedup_vs_fp16_like_for_like  same glue kernels (this repo's fused norms / attention) on both
        #                                  sides: what the INT8 GEMMs / convs themselves buy
        #   speedup_vs_fp16                against stock PyTorch FP16 ops throughout (the reference's
        #                                  comparison, kernels/README.md:108-110): glue gains included
        #   speedup_vs_fp16_dropin         the module swap alone (added below with the drop-in leg)
        out["speedup_vs_fp16"] = fp16["ms_per_step"] / ms
        if "fused_glue_ms_per_step" in fp16:
            out["speedup_vs_fp16_with_fused_glue"] = fp16["fused_glue_ms_per_step"] / ms
            out["speedup_vs_fp16_like_for_like"] = out["speedup_vs_fp16_with_fused_glue"]
        if "fp16" in memory:
            q = memory[kind]
            out["memory"]["saving_vs_fp16"] = {
                k: memory["fp16"][k] / q[k] for k in ("static_mb", "dynamic_mb", "peak_mb") if q[k]}
    if roof_stats:
        out["roofline"] = roofline_object(roof_stats, B, args.sweep_reps, args.px, args.tiny)
        if in_step:
            # The device kernel behind the dominant instantiation, inside one replay of the benchmarked graph.
            # (One device kernel serves several of this line's kernel names -- the GEMM+GEGLU epilogue is a
            # run-time branch of the plain Linear's instantiation -- so ops and time are summed over all of them.)
            r = out["roofline"]
            pat = kernel_name_pattern(r["kernel"])
            hits = [(n, t) for name, (n, t) in in_step.items() if pat in name]
            n_l, us_all = sum(n for n, _ in hits), sum(t for _, t in hits)
            ops_all = sum(v["ops"] / args.sweep_reps for k, v in roof_stats.items()
                          if not v.get("f16") and kernel_name_pattern(k) == pat)
            if n_l and us_all > 0:
                r["in_step_avg_launch_us"] = us_all / n_l
                r["in_step_launches"] = n_l
                r["frac_in_step"] = ops_all / (us_all * 1e-6) / 1e12 / INT8_MFMA_PEAK_TOPS
            r["in_step_kernel_time_ms"] = sum(t for _, t in in_step.values()) / 1e3
            r["in_step_kernels"] = sum(n for n, _ in in_step.values())
    if roof_stats:
        # whole step against the dense INT8 peak: every INT8 GEMM / conv op of one forward / the step time (the FP16
        # attention core's FLOPs are not INT8 work and are not counted)
        ops_step = out["roofline"]["all_igemm"]["int8_ops_per_step"]
        out["roofline"]["whole_step_int8_ops"] = ops_step
        out["roofline"]["whole_step_frac"] = ops_step / (ms * 1e-3) / 1e12 / INT8_MFMA_PEAK_TOPS
        rp = rocprof_kernel_avg(out["roofline"]["kernel"], B)
        if rp:
            out["roofline"].update(rp)
            out["roofline"]["frac_rocprof"] = (out["roofline"]["ops_per_launch"] / (rp["rocprof_avg_launch_us"] * 1e-6)
                                               / 1e12 / INT8_MFMA_PEAK_TOPS)
    if batch8:
        if "roofline" in batch8:
            ops8 = batch8["roofline"]["all_igemm"]["int8_ops_per_step"]
            batch8["whole_step_int8_ops"] = ops8
            batch8["whole_step_frac"] = ops8 / (batch8["ms_per_step"] * 1e-3) / 1e12 / INT8_MFMA_PEAK_TOPS
            rp8 = rocprof_kernel_avg(batch8["roofline"]["kernel"], 8)
            if rp8:
                batch8["roofline"].update(rp8)
                batch8["roofline"]["frac_rocprof"] = (batch8["roofline"]["achieved"] * batch8["roofline"]["avg_launch_us"]
                                                      / rp8["rocprof_avg_launch_us"] / INT8_MFMA_PEAK_TOPS)
        out["batch8"] = batch8
    if dropin:
        out.update(dropin)
        if fp16:
            out["speedup_vs_fp16_dropin"] = fp16["ms_per_step"] / dropin["dropin_unfused_ms_per_step"]
            out["speedup_vs_fp16_dropin_glue"] = fp16["ms_per_step"] / dropin["dropin_glue_ms_per_step"]
            out["speedup_vs_fp16_dropin_glue_torch_sdpa"] = fp16["ms_per_step"] / dropin["dropin_glue_torch_sdpa_ms_per_step"]
            out["speedup_vs_fp16_dropin_glue_own_quantize"] = fp16["ms_per_step"] / dropin["dropin_glue_own_quantize_ms_per_step"]
    if ln_in_gemm:
        out["ln_in_gemm"] = ln_in_gemm
    if world == 1:
        out["multi_gpu"] = ("unmeasured: no multi-GPU node was available to this build; RCCL has run under this "
                            "code at world size 1 only (tests/test_dist_gpu.py)")
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_fake_quant_baseline(args.cpu_seconds)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
