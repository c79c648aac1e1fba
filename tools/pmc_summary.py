"""Sum the counters of a rocprofv3 --pmc counter_collection.csv per kernel (all dispatches; counters are summed over
XCDs / SEs by the tool) and print per-dispatch means.  Usage: python tools/pmc_summary.py <counter_collection.csv> [filter]"""
import collections
import csv
import re
import sys


def main(path, flt=""):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        if flt and flt not in k:
            continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    for k in tot:
        n = len(disp[k])
        print(f"# kernel: {k[:120]}  ({n} dispatches; per-dispatch means)")
        for c in sorted(tot[k]):
            print(f"{c:32s} {tot[k][c] / n:18.0f}")
        t = {c: v / n for c, v in tot[k].items()}
        if "SQ_WAVE_CYCLES" in t and "SQ_ACTIVE_INST_ANY" in t:
            w = t["SQ_WAVE_CYCLES"]
            print(f"=> of SQ_WAVE_CYCLES: active {t['SQ_ACTIVE_INST_ANY']/w:.3f}, parked (WAIT_ANY) {t.get('SQ_WAIT_ANY',0)/w:.3f}, "
                  f"issue-stalled (WAIT_INST_ANY) {t.get('SQ_WAIT_INST_ANY',0)/w:.3f}, LDS-issue-stalled {t.get('SQ_WAIT_INST_LDS',0)/w:.3f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in t and "SQ_BUSY_CU_CYCLES" in t:
            print(f"=> MFMA pipe busy {t['SQ_VALU_MFMA_BUSY_CYCLES']/(4*t['SQ_BUSY_CU_CYCLES']):.3f}"
                  + (f"; VALU instructions per MFMA {t['SQ_INSTS_VALU']/t['SQ_INSTS_MFMA']:.2f}" if "SQ_INSTS_MFMA" in t and "SQ_INSTS_VALU" in t else "")
                  + (f"; LDS bank-conflict share {t['SQ_LDS_BANK_CONFLICT']/t['SQ_LDS_IDX_ACTIVE']:.3f}" if "SQ_LDS_IDX_ACTIVE" in t and "SQ_LDS_BANK_CONFLICT" in t else ""))
        print()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
