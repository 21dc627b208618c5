import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mixdq_amd._C as C
from oracle import oracle
from tests import detdata as dd
DEV = "cuda:0"
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
scal = lambda v: torch.tensor(float(v), dtype=torch.float32, device=DEV)
for (n, h, w_, c, k, has_bias, res_kind, tile) in [(1, 16, 16, 64, 72, False, "", 91), (2, 16, 16, 320, 320, True, "", 0), (1, 16, 32, 128, 80, True, "full", 90)]:
    x = dd.int8(901, (n, h, w_, c)); wt = dd.int8(902, (k, 3, 3, c)); scale = dd.f32(903, (k,), 1e-4, 6e-4)
    in_zp = -11.0
    bias = dd.f16(904, (k,), -1, 1) if has_bias else None
    wsum = wt.astype(np.float32).sum(axis=3, dtype=np.float32)
    args = (t(x).permute(0, 3, 1, 2), t(wt).permute(0, 3, 1, 2), t(scale), scal(1.0), scal(in_zp),
            t(scale), t(wsum.reshape(k, 1, 3, 3)), None, None if bias is None else t(bias), 1, 1)
    kw, add = {}, None
    if res_kind == "full":
        r = t(dd.normal_f16(905, (n, h, w_, k), 2.0)).permute(0, 3, 1, 2); kw, add = dict(_residual=r), r
    got = C.qconv2d_w8_a8_ohalf(*args, _cfg=tile, **kw)
    want = torch.from_numpy(oracle.qconv2d(x, wt, scale, wsum, in_zp, None, bias, 1, 1, C.FLAGS & 1)).to(DEV).permute(0, 3, 1, 2)
    if add is not None: want = want + add
    g4 = C.qconv2d_w8_a8_ohalf(*args, _cfg=4, **kw)
    for name, a in (("halo", got), ("cfg4", g4)):
        bad = (a != want).permute(0, 2, 3, 1)          # n h w c
        idx = bad.nonzero()
        print((n, h, w_, c, k, res_kind, tile), name, "mismatches", int(bad.sum()), "first", idx[:6].tolist(), "last", idx[-3:].tolist(),
              "vals", a.permute(0, 2, 3, 1)[bad][:4].tolist(), want.permute(0, 2, 3, 1)[bad][:4].tolist(), flush=True)
