"""Compile one .hip source for gfx950 and print a compact per-kernel resource table
(VGPRs / AGPRs / spills / LDS / occupancy) from hipcc's kernel-resource-usage remarks.

    python tools/kernel_resources.py mimo_amd/csrc/gemm_conv.hip [filter-substring]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "mimo_amd", "csrc"), "-c", src, "-o", "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        print(r.stderr[-4000:])
        sys.exit(1)
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = {"name": re.sub(r"\(anonymous namespace\)::|void |\(.*", "", name)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    print(f"{'kernel':44s} {'VGPR':>5s} {'AGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'LDS':>7s} {'occ':>3s}")
    for c in rows:
        if flt in c["name"]:
            print(f"{c['name'][:44]:44s} {c.get('VGPRs', -1):5d} {c.get('AGPRs', -1):5d} {c.get('VGPRs Spill', -1):6d} "
                  f"{c.get('SGPRs Spill', -1):6d} {c.get('ScratchSize', -1):7d} {c.get('LDS Size', -1):7d} {c.get('Occupancy', -1):3d}")


if __name__ == "__main__":
    main()
