#!/usr/bin/env python3
"""Static check of the AQ kernels' ISA (csrc/igemm_kernel.h, `AQ` instantiations).

The quantize-in-prologue GEMM requests its FP16 activation operand with inline-asm
`global_load_dwordx4` into registers that the compiler believes are written at once; the data
really arrives later, behind the counted `s_waitcnt vmcnt(N) ; AQWAIT` that is tied to the same
registers.  That is only sound if, between a request (`; AQLOAD <slot>`) and its wait
(`; AQWAIT <slot> regs...`), NO other instruction touches those registers (a copy or a spill there
would read registers whose load is still in flight).  This script proves that property on the
generated assembly of every AQ kernel:

  1. every `AQWAIT s` names exactly the registers the `AQLOAD s` requests of that kernel write
     (same physical registers in the prologue and in the loop: no PHI copies);
  2. walking the instructions in program order -- the K loop twice, so that requests issued late in
     the loop body are seen in flight at its top -- no instruction other than the slot's own wait
     mentions a register whose load is in flight;
  3. behind the loop every slot is retired by `s_waitcnt vmcnt(0)` before its registers are reused;
  4. no AQ kernel uses scratch memory (a spill of an in-flight register would be caught by 2; spills
     elsewhere are reported because they cost the prefetch its registers).

    python tools/check_aq_isa.py [igemm_aq.s]     (default: build it with hipcc -S into /tmp)
Exit status 0 = every AQ kernel passes.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_asm(out):
    sys.path.insert(0, ROOT)
    from mixdq_amd import build as B
    src = os.path.join(B.CSRC, "igemm_aq.hip")
    extra = [f for f in B.EXTRA.get("igemm_aq.hip", []) if not f.startswith("-save-temps")]
    cmd = [B._hipcc()] + B.FLAGS + extra + ["--cuda-device-only", "-S", "-o", out, src]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return out


REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def kernels(asm):
    """(name, [instruction lines], scratch bytes)"""
    lines = asm.split("\n")
    scratch = {}
    name = None
    for ln in lines:                      # .amdhsa metadata: private_segment_fixed_size per kernel
        m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", ln)
        if m:
            name = m.group(1)
        m = re.match(r"\s*\.amdhsa_private_segment_fixed_size\s+(\d+)", ln)
        if m and name:
            scratch[name] = int(m.group(1))
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\S*igemm_kernel\S*):", lines[i])
        if not m:
            i += 1
            continue
        body = []
        i += 1
        while i < len(lines) and "s_endpgm" not in lines[i]:
            body.append(lines[i])
            i += 1
        yield m.group(1), body, scratch.get(m.group(1), -1)


def check_kernel(name, body):
    errs = []
    ins = []                                   # (text, is_label)
    for ln in body:
        t = ln.strip()
        if not t or t.startswith(";") and "AQDONE" not in t or t.startswith(".") and not t.endswith(":"):
            continue
        ins.append(t)
    loads, waits = {}, {}
    for t in ins:
        m = re.search(r"; AQLOAD (\d+)", t)
        if m:
            dst = re.match(r"global_load_dwordx4\s+(v\[\d+:\d+\])", t)
            loads.setdefault(int(m.group(1)), []).append(dst.group(1) if dst else "?")
        m = re.search(r"; AQWAIT (\d+) (.*)$", t)
        if m:
            waits.setdefault(int(m.group(1)), []).append(m.group(2).split())
    if not loads:
        return ["no AQLOAD markers (not an AQ kernel?)"]
    slot_regs = {}
    for s, ws in waits.items():
        if any(w != ws[0] for w in ws):
            errs.append(f"slot {s}: waits name different registers: {ws}")
        slot_regs[s] = ws[0]
    for s, ls in loads.items():
        if s not in slot_regs:
            errs.append(f"slot {s}: loaded but never waited for")
            continue
        n = len(slot_regs[s])
        for g in range(0, len(ls), n):
            if ls[g:g + n] != slot_regs[s]:
                errs.append(f"slot {s}: a request writes {ls[g:g + n]}, its wait names {slot_regs[s]}")
    # program-order walk; the K loop = the outermost backward branch range that contains AQLOADs
    label_at = {t[:-1]: i for i, t in enumerate(ins) if t.endswith(":")}
    loop = None
    for i, t in enumerate(ins):
        m = re.match(r"s_cbranch\S*\s+(\S+)|s_branch\s+(\S+)", t)
        if m:
            tgt = label_at.get(m.group(1) or m.group(2))
            if tgt is not None and tgt < i and any("AQLOAD" in x for x in ins[tgt:i]):
                if loop is None or (tgt <= loop[0] and i >= loop[1]):
                    loop = (tgt, i)
    if loop is None:
        errs.append("K loop not found")
        return errs
    order = list(range(0, loop[1] + 1)) + list(range(loop[0], loop[1] + 1)) + list(range(loop[1] + 1, len(ins)))
    flying = {}                                # slot -> register set
    for idx in order:
        t = ins[idx]
        m = re.search(r"; AQLOAD (\d+)", t)
        if m:
            s = int(m.group(1))
            flying.setdefault(s, set()).update(regs_of(t.split(",")[0]))
            # the request's own address operands must not be in-flight registers of another slot
            for s2, rs in flying.items():
                if s2 != s and rs & regs_of(t.split(";")[0].split(",", 1)[1]):
                    errs.append(f"line {idx}: request for slot {s} reads in-flight registers of slot {s2}: {t}")
            continue
        m = re.search(r"; AQWAIT (\d+)", t)
        if m:
            flying.pop(int(m.group(1)), None)
            continue
        if re.match(r"s_waitcnt\s+vmcnt\(0\)", t):
            flying.clear()
            continue
        used = regs_of(t.split(";")[0])
        for s, rs in flying.items():
            if used & rs:
                errs.append(f"line {idx}: `{t}` touches v{sorted(used & rs)} while slot {s}'s load is in flight")
    return errs


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else build_asm("/tmp/igemm_aq_check.s")
    asm = open(path).read()
    bad = 0
    n = 0
    for name, body, scratch in kernels(asm):
        if "AQLOAD" not in "\n".join(body):
            continue
        n += 1
        errs = check_kernel(name, body)
        tag = re.search(r"igemm_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb\dELb\dELb(\d)ELi(\d+)ELi(\d+)", name)
        short = "igemm<%s,%s,%s,st%s,%sx%s,w4=%s,ks%s,mt%s>" % tag.groups() if tag else name
        status = "ok" if not errs else "FAIL"
        print(f"{short}: {status}  scratch={scratch} B")
        for e in errs[:8]:
            print("   ", e)
        bad += bool(errs)
    print(f"{n} AQ kernels checked, {bad} failed")
    return 1 if bad or n == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
