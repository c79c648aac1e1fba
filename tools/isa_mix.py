"""Instruction mix of a kernel's loops in a `hipcc -S --cuda-device-only` listing: per loop (backward branch), the count
of instructions by class and a crude issue-cycle budget per iteration (gfx950: MFMA 32x32x16 = 32 cycles of its SIMD's matrix
pipe, 16x16x32 = 16; transcendental VALU = quarter rate = 8 cycles per wave64, other VALU 2 — MI355X_MICROARCH.md "cycle
constants": v_fma_f32 wave64 = 2 cycles, SIMD-32 — DS / VMEM / SALU 1 issue slot of ~4 cycles each).
    python tools/isa_mix.py /tmp/k.s <name-substring> [min MFMAs per loop]"""
import re
import sys
from collections import Counter


def klass(op):
    op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if op.startswith("v_mfma"):
        return "mfma32" if "32x32" in op else "mfma16"
    if op in ("v_exp_f32", "v_rcp_f32", "v_log_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_exp_f16", "v_rcp_f16"):
        return "trans"
    if op.startswith("v_permlane") or op.startswith("v_readlane") or op.startswith("v_writelane") or "dpp" in op:
        return "lane"
    if op.startswith("v_cvt"):
        return "cvt"
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("scratch_") or op.startswith("flat_"):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop"):
        return "wait/nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


CYC = {"mfma32": 32, "mfma16": 16, "trans": 8, "valu": 2, "valu_pk": 4, "cvt": 2, "lane": 4}


def main(path, sub, min_mfma=8):
    lines = open(path).read().split("\n")
    s0 = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and sub in l)
    end = next(i for i in range(s0, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[s0:end]
    print(lines[s0].split(":")[0])
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    br = [(i, m.group(1)) for i, l in enumerate(body) for m in [re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)] if m]
    loops = sorted({(labels[t], i) for i, t in br if t in labels and labels[t] < i})
    for a, b in loops:
        ops = [m.group(1) for l in body[a:b + 1] for m in [re.match(r"\s+([a-z_0-9]+)", l)] if m and not l.strip().startswith(";")]
        c = Counter(klass(o) for o in ops)
        nm = c["mfma32"] + c["mfma16"]
        if nm < min_mfma:
            continue
        mf = c["mfma32"] * 32 + c["mfma16"] * 16
        va = sum(c[k] * CYC[k] for k in ("trans", "valu", "valu_pk", "cvt", "lane"))
        slots = c["lds"] + c["vmem"] + c["salu"] + c["wait/nop"]
        print(f"  loop {a}-{b}: {len(ops)} instructions; " + ", ".join(f"{k} {v}" for k, v in sorted(c.items())))
        print(f"    matrix-pipe cycles {mf}; VALU cycles {va} (transcendental {c['trans'] * 8}); other issue slots {slots} (x ~4 cycles = {slots * 4})")
        top = Counter(o for o in ops if klass(o) in ("valu", "valu_pk", "cvt", "lane", "trans")).most_common(12)
        print("    VALU by opcode: " + ", ".join(f"{o} {n}" for o, n in top))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 8)
