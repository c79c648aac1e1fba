"""Where do a kernel's scratch accesses sit?  Per function of a `hipcc -S --cuda-device-only` listing: its loops (backward
branches) with the barriers, MFMAs and scratch instructions inside each.
    python tools/isa_loops.py /tmp/k.s [name-substring]"""
import re
import sys


def main(path, sub=""):
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for si, s0 in enumerate(starts):
        name = lines[s0].split(":")[0]
        if sub not in name:
            continue
        end = next((i for i in range(s0, len(lines)) if lines[i].startswith(".Lfunc_end")), len(lines))
        body = lines[s0:end]
        bar = [i for i, l in enumerate(body) if "s_barrier" in l]
        scr = [i for i, l in enumerate(body) if re.match(r"\s+scratch_", l)]
        mf = [i for i, l in enumerate(body) if "v_mfma" in l]
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        br = [(i, m.group(1)) for i, l in enumerate(body) for m in [re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)] if m]
        loops = sorted({(labels[t], i) for i, t in br if t in labels and labels[t] < i})
        print(f"{name}: {len(body)} lines, {len(bar)} barriers, {len(mf)} MFMAs, {len(scr)} scratch instructions")
        for a, b in loops:
            cnt = lambda xs: sum(1 for x in xs if a <= x <= b)
            print(f"  loop {a}-{b}: barriers {cnt(bar)}, MFMAs {cnt(mf)}, scratch {cnt(scr)}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
