"""Scan the gfx950 assembly of csrc/knn.hip for the two hazards hipcc does not handle around the
hand-issued (inline asm) v_mfma_f32_4x4x1 instructions:
  * a VALU write of an MFMA's A/B register in the two instructions before it;
  * an MFMA whose accumulator input is the result of the MFMA issued immediately before it
    (a dependent MFMA needs two wait states: DESIGN.md section 4).
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
            total, bad = scan(out)
        print("%s: %d v_mfma_f32_4x4x1, %d suspicious" % (os.path.basename(f), total, len(bad)))
        for b in bad[:10]:
            print("   ", b)
        status |= 1 if bad else 0
    return status


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] or [os.path.join(ROOT, "3pu_pytorch_amd", "csrc", "knn.hip")]))
