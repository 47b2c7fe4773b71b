"""Scan the gfx950 assembly of the kernels for the hazards hipcc does not handle around inline asm:
  * (hand-issued v_mfma_f32_4x4x1 of csrc/knn.hip) a VALU write of an MFMA's A/B register in the two instructions
    before it; an MFMA whose accumulator input is the result of the MFMA issued immediately before it (a dependent
    MFMA needs two wait states: DESIGN.md section 4);
  * (any file) a VALU instruction INSIDE an inline-asm statement that reads a register a v_mfma wrote fewer than
    ASM_READ_SLOTS issue slots earlier: the compiler's hazard recognizer does not look into asm statements, so the
    wait states an MFMA result needs before a VALU read are not inserted (round 3: a `v_max_f32` in an asm statement
    read stale accumulators in one instantiation of the lane-per-point DenseEdgeConv kernel).  A register that a
    compiler-visible VALU instruction has overwritten since (round 5: `v_add_f32 v, x, |x|` = 2 relu into the
    accumulator's register) no longer counts.
Exit status 0 = clean.  Usage: python tools/check_mfma_hazards.py [file.hip ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan(asm_path):
    lines = [l.rstrip() for l in open(asm_path)]
    ins = [l.strip() for l in lines if l.startswith("\t") and not l.strip().startswith((";", "."))]
    total, bad = 0, []
    for k, l in enumerate(ins):
        if not l.startswith("v_mfma_f32_4x4x1"):
            continue
        total += 1
        ops = [t.strip() for t in l.split(None, 1)[1].split(",")]
        src = regs(ops[1]) | regs(ops[2])
        srcc = regs(ops[3]) if len(ops) > 3 else set()
        # walk back until two issue slots separate the producer from this MFMA (`s_nop n` fills n + 1 of them)
        slots, back = 0, 1
        while slots < 2 and k - back >= 0:
            pl = ins[k - back]
            m = re.match(r"s_nop\s+(\d+)", pl)
            if m:
                slots += int(m.group(1)) + 1
                back += 1
                continue
            if pl.startswith("v_") and not pl.startswith("v_mfma"):
                if regs(pl.split(None, 1)[1].split(",")[0].strip()) & src:
                    bad.append("A/B written just before use: %s -> %s" % (pl, l))
            if pl.startswith("v_mfma") and slots == 0:
                if regs(pl.split(None, 1)[1].split(",")[0].strip()) & srcc:
                    bad.append("dependent MFMA without wait states: %s -> %s" % (pl, l))
            slots += 1
            back += 1
    return total, bad


ASM_READ_SLOTS = 12          # (8-pass fp32 MFMAs need 11 wait states before a VALU read; 2-pass ones 5)


def scan_asm_reads(asm_path):
    """inline-asm VALU instructions that read a fresh MFMA result (see module doc)."""
    bad = []
    window = []                 # (slots ago, written registers) of recent MFMAs, newest last
    in_asm = False
    for raw in open(asm_path):
        l = raw.strip()
        if l.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if l.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not raw.startswith("\t") or l.startswith((";", ".")) or not l:
            if l.endswith(":"):
                window = []     # a label: control flow joins, nothing is known (loops are covered by their bodies)
            continue
        m = re.match(r"s_nop\s+(\d+)", l)
        step = int(m.group(1)) + 1 if m else 1
        if in_asm and l.startswith("v_") and not l.startswith("v_mfma"):
            ops = [t.strip() for t in l.split(None, 1)[1].split(",")] if " " in l else []
            srcs = set()
            for t in ops[1:]:
                srcs |= regs(t.split()[0]) if t else set()
            for age, dst, text in window:
                if age < ASM_READ_SLOTS and srcs & dst:
                    bad.append("asm VALU reads an MFMA result after %d slots: %s  <-  %s" % (age, l, text))
        if not in_asm and l.startswith("v_") and not l.startswith("v_mfma") and " " in l:
            # a compiler-visible VALU write replaces the MFMA result in its destination (the hazard recognizer has
            # covered that write): the register no longer holds a fresh accumulator
            over = regs(l.split(None, 1)[1].split(",")[0].strip())
            window = [(age, dst - over, text) for age, dst, text in window]
        window = [(age + step, dst, text) for age, dst, text in window if age + step < 64 and dst]
        if l.startswith("v_mfma"):
            window.append((0, regs(l.split(None, 1)[1].split(",")[0].strip()), l))
    return bad


def main(files):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    status = 0
    for f in files:
        with tempfile.TemporaryDirectory() as tmp:
            out = os.path.join(tmp, "k.s")
            subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                                   "-munsafe-fp-atomics", "-S", "--cuda-device-only", "-I",
                                   os.path.join(ROOT, "include"), "-o", out, f],
                                  stderr=subprocess.DEVNULL)
            total, bad = scan(out) if os.path.basename(f) == "knn.hip" else (0, [])
            bad = bad + scan_asm_reads(out)
        print("%s: %d hand-issued v_mfma_f32_4x4x1 checked, %d suspicious" % (os.path.basename(f), total, len(bad)))
        for b in bad[:10]:
            print("   ", b)
        status |= 1 if bad else 0
    return status


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] or [os.path.join(ROOT, "3pu_pytorch_amd", "csrc", "knn.hip")]))
