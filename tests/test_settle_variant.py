"""tools/settle_variant.py names the rounding variant of whatever operator module it is pointed at:
here an oracle-backed stand-in for the reference's `mixdq_extension._C`, once per variant."""
import subprocess
import sys
import os
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = textwrap.dedent('''
    import numpy as np, torch
    from oracle import oracle as O
    VARIANT = {variant}
    def _q(x, s_inv, zp):
        return torch.from_numpy(O.quantize(x.numpy(), float(s_inv), float(zp), VARIANT))
    quantize_per_tensor_to_int8 = quantize_per_tensor_to_int8_vectorized = _q
    def qlinear_w8_a8_ohalf(a, w, ws, a_s, a_zp, wsum, scale, bias0, bias=None):
        return torch.from_numpy(O.qlinear(a.numpy(), w.numpy(), bias0.numpy(), scale.numpy(),
                                          None if bias is None else bias.numpy(), VARIANT))
''')


@pytest.mark.parametrize("variant,word", [(0, "variant A"), (1, "variant B")])
def test_settle_variant_names_the_variant(tmp_path, oracle, variant, word):
    (tmp_path / "fake_ref_ext.py").write_text(STUB.format(variant=variant))
    env = dict(os.environ, PYTHONPATH=f"{tmp_path}:{ROOT}")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "settle_variant.py"),
                          "--ext", "fake_ref_ext", "--device", "cpu"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert f"VERDICT: {word}" in out.stdout, out.stdout
