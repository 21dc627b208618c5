#!/bin/bash
# Build the HIP library of another commit for same-box A/B runs: build/ab_<name>/libmixdq_hip.so
#   bash tools/ab_build.sh <commit> <name>;  MIXDQ_HIP_LIB=$PWD/build/ab_<name>/libmixdq_hip.so python bench.py ...
# The file lists (sources, headers, per-file flags) are taken from THAT commit's mixdq_amd/build.py.
# (mixdq_amd._C refuses a library of another ABI version: A/B across an ABI change needs that commit's
# Python too.)
set -e
cd "$(dirname "$0")/.."
c=$1; n=$2; d=build/ab_$n
rm -rf $d/src; mkdir -p $d/src/mixdq_amd/csrc $d/src/include
git show $c:mixdq_amd/build.py > $d/src/build_py.py
python3 - "$c" "$d" <<'PY'
import importlib.util, os, subprocess, sys
c, d = sys.argv[1:3]
spec = importlib.util.spec_from_file_location("build_py", os.path.join(d, "src", "build_py.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
csrc = os.path.join(d, "src", "mixdq_amd", "csrc")
for f in list(b.SOURCES) + list(b.HEADERS):
    rel = os.path.normpath(os.path.join("mixdq_amd/csrc", f))
    out = os.path.normpath(os.path.join(csrc, f))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    open(out, "wb").write(subprocess.check_output(["git", "show", f"{c}:{rel}"]))
procs, objs = [], []
for src in b.SOURCES:
    obj = os.path.join(d, src.replace(".hip", ".o")); objs.append(obj)
    flags = [f for f in b.FLAGS if f != "-Wall"] + list(getattr(b, "EXTRA", {}).get(src, []))
    procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc"] + flags + ["-c", "-o", obj, os.path.join(csrc, src)]))
assert all(p.wait() == 0 for p in procs), "compile failed"
# the AQ kernels park in-flight loads in registers the compiler believes written: the static check of the build
# (mixdq_amd/build.py _check_aq_isa) on THIS library's assembly too, same flags, -S instead of -c
if "igemm_aq.hip" in b.SOURCES and os.path.exists("tools/check_aq_isa.py"):
    flags = [f for f in b.FLAGS if f != "-Wall"] + [f for f in getattr(b, "EXTRA", {}).get("igemm_aq.hip", []) if not f.startswith("-save-temps")]
    asm = os.path.join(d, "igemm_aq.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["--cuda-device-only", "-S", "-o", asm, os.path.join(csrc, "igemm_aq.hip")],
                          stderr=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, "tools/check_aq_isa.py", asm], capture_output=True, text=True)
    assert r.returncode == 0, "AQ ISA check failed for this library:\n" + r.stdout[-2000:]
    os.remove(asm)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                       os.path.join(d, "libmixdq_hip.so")] + objs)
for o in objs:
    os.remove(o)
PY
rm -rf $d/src
echo $d/libmixdq_hip.so
