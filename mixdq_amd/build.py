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
HEADERS = ["common.h", "attn_core.h", "iconv.h", "igemm_kernel.h", "igemm_pp.h", "f16in_table.h", os.path.join("..", "..", "include", "mixdq_hip.h"),
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


def csrc_sha16() -> str:
    """sha256 (first 16 hex digits) over the kernel sources: embedded in the library at build time
    (mixdq_build_csrc_sha16) so that a run can say which sources the LOADED library was built from --
    hashing the tree on disk says nothing about a stale .so or one named by MIXDQ_HIP_LIB."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(CSRC, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def _build_info_obj(force: bool) -> str:
    """A host-only object that carries the hash of the sources being compiled (rebuilt whenever it changes)."""
    src = os.path.join(OBJ, "build_info.cpp")
    obj = os.path.join(OBJ, "build_info.o")
    text = ('extern "C" __attribute__((visibility("default"))) const char* mixdq_build_csrc_sha16() '
            '{ return "%s"; }\n' % csrc_sha16())
    if force or not os.path.exists(src) or open(src).read() != text or not os.path.exists(obj):
        with open(src, "w") as f:
            f.write(text)
        subprocess.check_call(["g++", "-O1", "-fPIC", "-c", "-o", obj, src])
    return obj


def _aq_asm_cmd(asm: str):
    """The device assembly of igemm_aq.hip under EXACTLY the flags of its object (same compiler, same options,
    -S instead of -c: the same code generation; `-save-temps`, which would hand over the object's own .s, is not an
    option -- it switches hipcc to its non-integrated pipeline, which generates OTHER code: 48 bytes of scratch in
    kernels that have none)."""
    src = os.path.join(CSRC, "igemm_aq.hip")
    return [_hipcc()] + FLAGS + EXTRA.get("igemm_aq.hip", []) + ["--cuda-device-only", "-S", "-o", asm, src]


def _check_aq_isa(asm: str, obj: str) -> None:
    """tools/check_aq_isa.py on that assembly: the quantize-in-prologue kernels park in-flight loads in registers
    the compiler believes written, which is only sound if nothing touches them before their tied wait -- a
    toolchain or flag change that breaks that must fail the BUILD, not ship."""
    tool = os.path.join(os.path.dirname(PKG), "tools", "check_aq_isa.py")
    if not os.path.exists(tool):          # (a source tree without tools/: nothing to run)
        return
    r = subprocess.run([sys.executable, tool, asm], capture_output=True, text=True)
    if r.returncode != 0:
        if os.path.exists(obj):
            os.remove(obj)
        raise RuntimeError("igemm_aq.hip: the AQ kernels touch a register whose load is in flight\n" + r.stdout[-4000:])


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
            procs.append((cmd, subprocess.Popen(cmd), src, obj))
            if src == "igemm_aq.hip":              # its assembly, beside it (see _aq_asm_cmd)
                asm_cmd = _aq_asm_cmd(os.path.join(OBJ, "igemm_aq.s"))
                procs.append((asm_cmd, subprocess.Popen(asm_cmd, stderr=subprocess.DEVNULL), "igemm_aq.s", obj))
        for cmd, pr, src, obj in procs:            # the files compile in parallel
            if pr.wait() != 0:
                raise subprocess.CalledProcessError(pr.returncode, cmd)
        for cmd, pr, src, obj in procs:
            if src == "igemm_aq.s":
                _check_aq_isa(os.path.join(OBJ, "igemm_aq.s"), obj)
        objs.append(_build_info_obj(force))
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
