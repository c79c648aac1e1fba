"""MFMA-pipe utilisation per kernel FAMILY over one rocprofv3 --pmc pass of tools/profile_forward.py:
  python tools/pmc_family.py <counter_collection.csv>
Counters needed in the pass: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY."""
import collections
import csv
import re
import sys


def family(k):
    k = re.sub(r"\(anonymous namespace\)::", "", k)
    for name in ("hconv_kernel", "gemm8_kernel", "thin_conv_kernel", "ff_fused_kernel", "ff4_kernel", "gemm_stream_kernel", "gemm_dense_persist_kernel", "splitk_reduce_kernel",
                 "attn40_kernel", "temporal_attn2_kernel", "temporal_attn_kernel",
                 "attn_kernel", "gn_apply_kernel", "gn_stats", "layer_norm_kernel"):
        if name in k:
            return name
    m = re.search(r"gemm_kernel<\d+, \d+, (\d+)", k)
    if m:
        return "gemm_kernel (dense)" if m.group(1) == "0" else "gemm_kernel (3x3 conv)"
    return None


def main(path):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        f = family(r["Kernel_Name"])
        if f:
            tot[f][r["Counter_Name"]] += float(r["Counter_Value"])
            n[f].add(r["Dispatch_Id"])
    extra = sorted({c for t in tot.values() for c in t} - {"SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
                                                          "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"})
    if extra:  # any other counters of the pass: per busy CU cycle
        print(f"{'family':28s} {'dispatches':>10s} " + " ".join(f"{c[-22:]:>22s}" for c in extra) + "   (per SQ_BUSY_CU_CYCLES)")
        for f, t in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", 0)):
            b = t.get("SQ_BUSY_CU_CYCLES", 0) or float("nan")
            print(f"{f:28s} {len(n[f]):10d} " + " ".join(f"{t.get(c, 0) / b:22.4f}" for c in extra))
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in {c for t in tot.values() for c in t}:
            return
    print(f"{'family':28s} {'dispatches':>10s} {'MFMA busy':>10s} {'active':>8s} {'parked':>8s} {'stalled':>8s}")
    for f, t in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", 0)):
        w = t.get("SQ_WAVE_CYCLES", 0) or 1
        busy = t.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * t["SQ_BUSY_CU_CYCLES"]) if t.get("SQ_BUSY_CU_CYCLES") else float("nan")
        print(f"{f:28s} {len(n[f]):10d} {busy:10.3f} {t.get('SQ_ACTIVE_INST_ANY',0)/w:8.3f} {t.get('SQ_WAIT_ANY',0)/w:8.3f} {t.get('SQ_WAIT_INST_ANY',0)/w:8.3f}")


if __name__ == "__main__":
    main(sys.argv[1])
