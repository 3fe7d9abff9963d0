#!/usr/bin/env python3
"""Lint of the compiled kernels for the hazard inline-asm MFMAs are exposed to: hipcc pads "vector instruction writes a VGPR -> MFMA
reads it as an operand" for its OWN MFMAs, but an `asm volatile("v_mfma...")` is an opaque instruction to it -- if register
allocation makes it assemble an operand tuple with a v_mov (or any VALU) right in front of the asm, the MFMA may read the OLD register
(timing dependent: run-to-run differences in the last bits).  For every csrc/*.hip: compile to ISA, and for every v_mfma inside an
#ASMSTART / #ASMEND block report VALU writes to its A / B / C operand registers within the last WAIT wait states -- and, the other
direction, any vector / LDS / memory instruction that reads such an MFMA's result within RESULT_WAIT wait states (what fs_mfma_settle
is there for) -- and any MFMA (the compiler's own included) that takes such a result as its A or B operand that soon (an MFMA may
chain on a predecessor's result as its ACCUMULATOR back to back; as a multiplicand it may not).  The report counts the asm MFMAs it
inspected per kernel, so that "none found" can be told from "none looked at".

    python tools/check_mfma_hazard.py [file.hip ...]        exit code 1 if anything is found
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "fly_bproject_amd", "csrc")
WAIT = 4
RESULT_WAIT = 11          # an 8-pass MFMA's result -> a vector / memory instruction reading it


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def demangle_short(sym):
    """`mlp_fused_step_kernel<0>`-style name of a mangled kernel symbol."""
    m = re.findall(r"\d+([a-z][a-z0-9_]*kernel)(I(?:L[ib]\d+E)+E)?", sym)
    if not m:
        return sym[:60]
    name, targs = m[-1]
    if targs:
        name += "<" + ", ".join(re.findall(r"L[ib](\d+)E", targs)) + ">"
    return name


def check(path):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-I" + os.path.join(REPO, "include"),
                               "-I" + CSRC, "-Wno-unused-function", "-S", "--cuda-device-only", "-o", out, path], stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    found = []
    counts = {}          # kernel -> asm MFMAs inspected
    in_asm = False
    hist = []            # (wait states ago accumulates, written regs, text) of recent instructions
    res = []             # (wait states ago, result regs, text) of recent asm MFMAs: a non-MFMA reader needs RESULT_WAIT wait states
    func = "?"
    for ln, raw in enumerate(lines, 1):
        s = raw.strip()
        m = re.match(r"^(_Z[\w$]+):", raw)          # a kernel's label (mangled; a `; @name` comment may follow)
        if m:
            func = demangle_short(m.group(1))
        if "#ASMSTART" in s:
            in_asm = True
            continue
        if "#ASMEND" in s:
            in_asm = False
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        op = s.split()[0]
        if op == "s_nop":
            n = int(s.split()[1]) + 1
            hist = [(w + n, r, t) for (w, r, t) in hist]
            res = [(w + n, r, t) for (w, r, t) in res]
            continue
        args = s[len(op):].split(",")
        if op.startswith("v_mfma") and in_asm:
            counts[func] = counts.get(func, 0) + 1
            src = regs(args[1]) | regs(args[2]) | (regs(args[3]) if len(args) > 3 else set())      # A, B and the accumulator input
            for w, r, t in hist:
                if w < WAIT and r & src:
                    found.append((os.path.basename(path), func, ln, t, s))
        if op.startswith("v_mfma"):     # any MFMA multiplying by an asm MFMA's fresh result (A / B operand; chaining on C is fine)
            ab = regs(args[1]) | regs(args[2])
            for w, r, t in res:
                if w < RESULT_WAIT and r & ab:
                    found.append((os.path.basename(path), func, ln, t, s))
        # the other direction: something that is not an MFMA reads an asm MFMA's result before it is written (fs_mfma_settle's job)
        if not op.startswith("v_mfma") and (op.startswith("v_") or op.startswith("ds_") or op.startswith("global_") or op.startswith("scratch_")):
            srcs = set()
            for a in (args if (op.startswith("ds_write") or op.startswith("global_store") or op.startswith("scratch_store")) else args[1:]):
                for tok in re.findall(r"v\[\d+:\d+\]|v\d+", a):
                    srcs |= regs(tok)
            for w, r, t in res:
                if w < RESULT_WAIT and r & srcs:
                    found.append((os.path.basename(path), func, ln, t, s))
        res = [(w + 1, r, t) for (w, r, t) in res if w + 1 < RESULT_WAIT + 2]
        if op.startswith("v_mfma") and in_asm:
            res.append((0, regs(args[0]), s))
        # every instruction is one wait state
        hist = [(w + 1, r, t) for (w, r, t) in hist if w + 1 < WAIT + 2]
        if op.startswith("v_") and not op.startswith("v_mfma") and not op.startswith("v_cmp"):
            hist.append((0, regs(args[0]), s))
    return found, counts


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    bad, looked = [], 0
    for f in files:
        found, counts = check(f)
        bad += found
        for k, n in sorted(counts.items()):
            print("%s %s: %d asm MFMAs inspected" % (os.path.basename(f), k, n))
            looked += n
    for b in bad:
        print("%s %s line %d: `%s` in front of `%s`" % b)
    print("%d hazard(s) among %d asm MFMAs in %d file(s)" % (len(bad), looked, len(files)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
