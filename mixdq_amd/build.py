"""Build libmixdq_hip.so (the C-ABI library, include/mixdq_hip.h) in-tree with hipcc for gfx950.

    python -m mixdq_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the gpurun snapshot.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libmixdq_hip.so")
SOURCES = ["quantize.hip", "igemm.hip", "igemm_aq.hip", "igemm_ln.hip", "iconv.hip", "fused_norm.hip", "attention.hip"]
HEADERS = ["common.h", "attn_core.h", "iconv.h", "igemm_kernel.h", os.path.join("..", "..", "include", "mixdq_hip.h"),
           os.path.join("..", "..", "include", "mixdq_math.h")]
# -ffp-contract=off: every fused multiply-add in the arithmetic specification is written
# explicitly (__builtin_fmaf); the compiler must not introduce others.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]
# Per-file extras.  attention.hip: keep the MFMA accumulators in VGPRs — the softmax works on them
# between the two products, and the default AGPR form costs ~190 v_accvgpr moves per K/V tile.
# igemm.hip: the dispatcher preloads the first 14 dwords of the kernel arguments into scalar registers
# (csrc/igemm.hip MIXDQ_KP: the operands the prologue DMAs depend on are leading scalar arguments), so the
# first memory request of a launch does not wait for a scalar-cache round trip: (1024, 1280, 1280) in a
# dependent chain 5.95 -> 5.79 us, batch-1 step 11.60 -> 11.41 ms (same box, tools/kp_build.sh A/B).
# Not on the other files: a kernel that gains nothing from it pays for the preload at every wave
# launch (quantize on 8 elements: 1.58 -> 1.71 us).
EXTRA = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
         "igemm.hip": ["-mllvm", "-amdgpu-kernarg-preload-count=14"],
         "igemm_aq.hip": ["-mllvm", "-amdgpu-kernarg-preload-count=14"],
         "igemm_ln.hip": ["-mllvm", "-amdgpu-kernarg-preload-count=14"]}
OBJ = os.path.join(PKG, "_obj")


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    return "hipcc"


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build():
        os.makedirs(OBJ, exist_ok=True)
        hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
        hdr_t = max(hdr_t, os.path.getmtime(os.path.abspath(__file__)))
        procs, objs = [], []
        for src in SOURCES:
            path = os.path.join(CSRC, src)
            obj = os.path.join(OBJ, src.replace(".hip", ".o"))
            objs.append(obj)
            if (not force and os.path.exists(obj)
                    and os.path.getmtime(obj) > max(hdr_t, os.path.getmtime(path))):
                continue
            cmd = [_hipcc()] + FLAGS + EXTRA.get(src, []) + ["-c", "-o", obj, path]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
        for cmd, pr in procs:                      # the files compile in parallel
            if pr.wait() != 0:
                raise subprocess.CalledProcessError(pr.returncode, cmd)
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
