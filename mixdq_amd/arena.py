"""One allocation for the static data of a converted network (`pack_static_`).

After `quantize_unet` the weights live where PyTorch's caching allocator put them while the FP16 modules were being
replaced one by one: ~2 700 tensors (INT8 weights, per-channel epilogue vectors, FP16 parameters) carved out of a few
hundred 2 / 20 MB segments, interleaved with whatever was freed in between.  A batch-1 step reads every one of them
exactly once, cold, and the first staged K-tile of a GEMM waits for that read (~2.5 us from the
request to the first landed stage).  `pack_static_` moves them, in module order, into ONE buffer and re-points the
tensors at it (same tensor objects, same names, same shapes and strides: state_dict, hooks, packs of
mixdq_amd.unet._pack_rows and weak references all stay valid), so that consecutive launches read consecutive memory and
the address translation of the whole set is as few, as large fragments as the driver can map.

Host logic only: no kernel, nothing the oracle has to restate -- the bytes every launch reads are the same bytes.
"""
from __future__ import annotations

import torch

ALIGN = 256          # bytes: every moved storage starts on a 256-byte line of the arena (LDS-DMA wants 16)


def _static_tensors(root):
    """Every tensor the modules of `root` own: parameters, buffers (views of a row pack included) and the
    row packs themselves (`_qkv` / `_kv` dictionaries of mixdq_amd.unet), in module order, each once."""
    seen, out = set(), []

    def add(t):
        if torch.is_tensor(t) and id(t) not in seen and t.numel():
            seen.add(id(t))
            out.append(t)

    for m in root.modules():
        for t in m._parameters.values():
            add(t)
        for t in m._buffers.values():
            add(t)
        for v in m.__dict__.values():
            if isinstance(v, dict) and "layers" in v and "names" in v:      # a row pack
                for t in v.values():
                    add(t)
    return out


def pack_static_(root, device=None) -> dict:
    """Move the static tensors of `root` on `device` (default: the device of its first tensor) into one
    allocation, storage by storage (tensors that alias one storage keep aliasing it).  Returns
    {"bytes", "storages", "tensors"}; the arena is kept alive by the tensors that now view it.  Call it before
    a hipGraph of the network is captured (a captured graph holds the old addresses), and again after anything
    that re-allocates the tensors (`.to()`, load_state_dict with assign=True, a second quantize_unet)."""
    tensors = _static_tensors(root)
    if device is None and tensors:
        device = tensors[0].device
    tensors = [t for t in tensors if t.device == device]
    groups, order = {}, []
    for t in tensors:
        key = t.untyped_storage().data_ptr()
        if key not in groups:
            groups[key] = []
            order.append(key)
        groups[key].append(t)
    offs, total = {}, 0
    for key in order:
        offs[key] = total
        total += -(-groups[key][0].untyped_storage().nbytes() // ALIGN) * ALIGN
    if total == 0:
        return dict(bytes=0, storages=0, tensors=0)
    arena = torch.empty(total, dtype=torch.uint8, device=device)
    dst = arena.untyped_storage()
    with torch.no_grad():
        for key in order:
            src = groups[key][0].untyped_storage()
            n = src.nbytes()
            raw = torch.empty(0, dtype=torch.uint8, device=device).set_(src, 0, (n,), (1,))
            arena[offs[key]:offs[key] + n].copy_(raw)
        if arena.is_cuda:
            torch.cuda.synchronize(device)
        for key in order:
            for t in groups[key]:
                es = t.element_size()
                t.set_(dst, offs[key] // es + t.storage_offset(), t.size(), t.stride())
    return dict(bytes=total, storages=len(order), tensors=len(tensors))
