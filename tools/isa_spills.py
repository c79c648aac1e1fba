"""Scratch (spill) instructions of one kernel in a `hipcc -S` listing, by barrier-delimited segment, with the first few shown.
    python tools/isa_spills.py /tmp/k.s <name-substring>"""
import bisect
import re
import sys
from collections import Counter


def main(path, sub):
    lines = open(path).read().split("\n")
    s0 = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and sub in l)
    end = next(i for i in range(s0, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[s0:end]
    bar = [i for i, l in enumerate(body) if "s_barrier" in l]
    st = [i for i, l in enumerate(body) if re.match(r"\s+scratch_store", l)]
    ld = [i for i, l in enumerate(body) if re.match(r"\s+scratch_load", l)]
    seg = lambda i: bisect.bisect(bar, i)
    print("stores by segment", sorted(Counter(seg(i) for i in st).items()))
    print("loads by segment ", sorted(Counter(seg(i) for i in ld).items()))
    for i in st[:6] + ld[:6]:
        print(i, body[i].strip())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
