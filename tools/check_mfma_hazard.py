"""Dev / CI tool: find MFMA results that a vector instruction reads too early ACROSS A BRANCH in hipcc's gfx950 output.

Why (round 6, the cause of round 5's "racy" 64x64 part-wise backward): on gfx950 the distance between an MFMA and a VALU /
LDS / VMEM instruction that reads its destination registers is the compiler's job (s_nop padding: 4-pass XDL 16x16x32 needs
7 wait states, 8-pass 32x32x16 needs 11; LLVM GCNHazardRecognizer).  hipcc pads along the LAYOUT order of the blocks; a
conditional branch that jumps over the padded block lands on code that was never checked against the MFMAs issued just
before the branch.  In the persistent-tile build of lora_gemm.hip `if (p.bias != nullptr)` sat between the last rank-step MFMA
and `v_cvt_pk_f16_f32 v5, v4, v5` — two instructions after the branch target, reading the MFMA's own destination: rows 16–31 /
columns 2–3 of the tile came out as whatever the registers held, differently on every run.

usage: python tools/check_mfma_hazard.py file.s [...]      (hipcc -S --cuda-device-only output)   exit code 1 on a finding
"""
import re
import sys

NEED = {"16x16x32": 7, "16x16x16": 7, "32x32x16": 11, "32x32x8": 11, "16x16x4": 7, "32x32x2": 11, "4x4x4": 5}
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def states(ins):
    m = re.match(r"s_nop (\d+)", ins)
    return int(m.group(1)) + 1 if m else 1


def parse(path):
    kernels, cur, name = {}, None, None
    for line in open(path):
        line = line.rstrip("\n")
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, cur = m.group(1), []
            kernels[name] = cur
            continue
        if cur is None:
            continue
        s = line.split(";")[0].strip()
        if not s or s.startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", s or ""):
                cur.append(("label", s[:-1]))
            elif s.startswith(".Lfunc_end"):
                cur = None
            continue
        cur.append(("ins", s))
    return kernels


def check(kernels):
    findings = []
    for name, body in kernels.items():
        labels = {t: i for i, (k, t) in enumerate(body) if k == "label"}
        for i, (k, t) in enumerate(body):
            if k != "ins" or not re.match(r"s_c?branch", t):
                continue
            target = t.split()[-1]
            if target not in labels:
                continue
            # MFMAs issued shortly before the branch (same block), with the wait states already elapsed after each
            pending, elapsed, j = [], 0, i - 1
            while j >= 0 and body[j][0] == "ins" and elapsed < 12:
                ins = body[j][1]
                m = re.match(r"v_mfma_\w+?_(\d+x\d+x\d+)\w*\s+(\S+),", ins)
                if m and m.group(2).startswith("v"):
                    need = NEED.get(m.group(1), 11)
                    if elapsed < need:
                        pending.append((regs(m.group(2)), need - elapsed, ins))
                elapsed += states(ins)
                j -= 1
            if not pending:
                continue
            # walk the TARGET block: a reader of a pending destination within the remaining wait states is a finding
            gone, j = 1, labels[target] + 1  # (the branch itself is one state)
            while j < len(body) and body[j][0] == "ins" and pending:
                ins = body[j][1]
                ops = ins.split(None, 1)
                src = ops[1] if len(ops) > 1 else ""
                if not ins.startswith(("v_mfma", "s_")):  # MFMA→MFMA chains have their own (shorter) rules; scalar ops read no VGPR
                    parts = src.split(",", 1)
                    reads = regs(parts[1] if len(parts) > 1 and not ins.startswith(("ds_write", "global_store", "buffer_store", "scratch_store")) else src)
                    for dst, left, mf in pending:
                        if reads & dst and gone < left:
                            findings.append((name, t, mf, ins, left - gone))
                gone += states(ins)
                pending = [(d, l, m) for d, l, m in pending if gone < l]
                if re.match(r"s_c?branch|s_endpgm", ins):
                    break
                j += 1
    return findings


if __name__ == "__main__":
    bad = []
    for path in sys.argv[1:]:
        f = check(parse(path))
        for name, br, mf, ins, short in f:
            print(f"{path}: {name[:90]}\n    after `{br}`: `{ins}` reads the result of `{mf}` {short} wait state(s) early")
        bad += f
    print(f"{len(bad)} finding(s)")
    sys.exit(1 if bad else 0)
