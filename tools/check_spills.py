#!/usr/bin/env python3
"""Build-time check of the device assembly hipcc generated for a kernel file (run by ppo_amd/csrc/Makefile on the
`-save-temps` output of the very compile that produces the object; the build fails on a finding).

Why: hipcc 7.2 (ROCm 7.2.0) can lose part of a spilled register tuple.  Under register pressure it splits the spill of
a 128-bit value -- three dwords to scratch, the fourth to a free AGPR (tagged `; Reload Reuse` in the assembly) -- and
the reload restores the three dwords only.  Seen in k_rollout<CartPole, three-piece split nets, FULL> of round 3: a
layer-1 weight fragment came back as a[220:222] and the MFMA read a[220:223].  The dword that was lost multiplies
input slots that are zero, so whatever the register held dropped out -- unless it read as a bf16 Inf / NaN, which
turned every output of the wave into NaN: 2-7 of a launch's first 1024 waves, depending on what the kernel before
had left in that register.  No test can be relied on to catch that; the assembly can.

Checks, per kernel:
  1. a register written by spill code (`; Reload Reuse` / `Folded Reload`) and never read afterwards;
  2. a matrix-core instruction reading a register that nothing before it (in layout order) has written;
  3. a function call inside a kernel (s_swappc): every device function here is meant to be inlined -- when hipcc
     outlined the rollout kernel's second network pass (a lambda with three call sites), the callee got the
     register-resident network through scratch and returned wrong values for half the wave.
usage: check_spills.py file.s [file.s ...]      exit status 1 on a finding."""
import re
import sys

KERNEL = re.compile(r"^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$")
NO_DEST = ("s_", "buffer_store", "global_store", "flat_store", "ds_write", "scratch_store", "v_cmp", "v_nop",
           "global_atomic", "buffer_atomic", "ds_add", "ds_max", "ds_min", "ds_or", "ds_and")
TWO_DEST = ("v_div_scale", "v_mad_u64_u32", "v_mad_i64_i32", "v_add_co", "v_sub_co", "v_subrev_co", "v_addc_co",
            "v_subb_co", "v_subbrev_co")


def regs(text):
    out = []
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]", text):
        out += [m.group(1) + str(i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    for m in re.finditer(r"(?<![\w\[:])([va])(\d+)\b", text):
        out.append(m.group(1) + m.group(2))
    return out


def instructions(body):
    for ln, raw in body:
        text = raw.split(";")[0].strip()
        if not text or text.startswith(".") or text.endswith(":") or "ASMSTART" in raw or "ASMEND" in raw:
            continue
        parts = text.split(None, 1)
        if len(parts) < 2:
            continue
        op = parts[0]
        ops = [x.strip() for x in re.split(r",(?![^\[]*\])", parts[1])]
        if op.startswith("v_permlane32_swap") or op.startswith("v_permlane16_swap") or op.startswith("v_swap"):
            dests, srcs = ops, ops
        elif op.startswith(NO_DEST):
            dests, srcs = [], ops
        else:
            n = 2 if op.startswith(TWO_DEST) else 1
            dests, srcs = ops[:n], ops[n:]
        yield ln, raw, op, dests, srcs


def check_kernel(name, body):
    findings = []
    written, read = set(), set()
    spilled = {}                                   # register -> line of the spill code that wrote it
    for ln, raw, op, dests, srcs in instructions(body):
        if op.startswith("s_swappc"):
            findings.append(f"{name}: line {ln}: function call inside the kernel: {raw.strip()}")
        for s in srcs:
            for r in regs(s):
                read.add(r)
                if op.startswith("v_mfma") and r not in written:
                    findings.append(f"{name}: line {ln}: {op} reads {r}, which nothing before it writes: {raw.strip()}")
        for d in dests:
            for r in regs(d):
                written.add(r)
                if "Reload Reuse" in raw or "Folded Reload" in raw:
                    spilled.setdefault(r, ln)
    for r, ln in spilled.items():
        if r not in read:                          # (anywhere in the kernel: a loop may read it on its next trip)
            findings.append(f"{name}: line {ln}: spill code writes {r} and nothing in the kernel reads it")
    return findings


def check_file(path):
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if KERNEL.match(l) and not l.startswith(".")]
    findings, kernels = [], 0
    for k, i in enumerate(starts):
        end = starts[k + 1] if k + 1 < len(starts) else len(lines)
        body = [(j + 1, lines[j]) for j in range(i + 1, end)]
        if not any("s_endpgm" in l for _, l in body):
            continue
        kernels += 1
        findings += check_kernel(lines[i].split(":")[0][:90], body)
    return kernels, findings


def main():
    bad = 0
    for path in sys.argv[1:]:
        kernels, findings = check_file(path)
        for f in findings:
            print("check_spills:", f)
        bad += len(findings)
        print(f"check_spills: {path}: {kernels} kernels, {len(findings)} findings")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
