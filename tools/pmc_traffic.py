"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/profile_forward.py into the per-launch HBM-side
traffic of the gemm_kernel family (the `roofline.traffic` figure of bench.py).

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> > profiles/<name>.json

Units and corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE
counts a 128-byte request as 64 bytes for wide coalesced reads, so it is doubled; WRITE_SIZE is uncalibrated and reported as is.
Infinity-Cache hits are included (fabric-side counters).
(Until the end of round 5 the family match missed `gemm8_kernel` — 105 of the 275 launches bench.py brackets per forward — and, for
the last pass of that round, the new `ff4_kernel`: the committed r4 / r5 summaries are means over the OTHER launches of the family.)"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    tot = collections.defaultdict(float)
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        fam = "gemm_kernel" if ("gemm_kernel" in k or "gemm8_kernel" in k or "gemm_dense_persist_kernel" in k or "gemm_stream_kernel" in k or "splitk_reduce" in k
                                or "hconv_kernel" in k or "ff_fused_kernel" in k or "ff4_kernel" in k) else \
              "attn_kernel" if ("attn_kernel" in k or "attn40_kernel" in k) and "temporal" not in k else None
        if fam:
            tot[fam] += float(r["Counter_Value"])
            n[fam].add(r["Dispatch_Id"])
    return {f: (tot[f], len(n[f])) for f in tot}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for fam in fetch:
        f_kb, nf = fetch[fam]
        w_kb, nw = write.get(fam, (0.0, 1))
        out[fam] = dict(launches_profiled=nf,
                        fetch_bytes_per_launch=2.0 * f_kb * 1024 / nf,     # x2: gfx950 FETCH_SIZE correction
                        write_bytes_per_launch=w_kb * 1024 / max(nw, 1),
                        note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over tools/profile_forward.py; "
                             "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950), WRITE_SIZE as reported; fabric-side (MALL hits included)")
        out[fam]["traffic_bytes_per_launch"] = out[fam]["fetch_bytes_per_launch"] + out[fam]["write_bytes_per_launch"]
    import hashlib
    import os
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mimo_amd", "libmimo_hip.so")
    out["library_sha256_16"] = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]  # the build the counters were taken on
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
