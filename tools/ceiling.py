"""The MFMA ceiling of THIS chip under THIS precision policy (round-5 verdict item 6), three rungs:
  1. nothing but MFMAs in flight (tools/probes/mfma_ceiling_probe.hip: register operands, random data vs zeros, 1 | 2 waves per SIMD),
  2. a best-case real GEMM: 8192^3 half x half -> half through the shipped 8-phase kernel (gemm8_kernel), random data,
  3. the same with the fp32-out + fp32-residual epilogue the residual stream imposes on the token GEMMs.
  python tools/ceiling.py > profiles/r5_mfma_ceiling.txt       (build the probe first, see its header)"""
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimo_amd import ops  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    probe = os.path.join(ROOT, "tools", "_ab", "mfma_ceiling_probe")
    if os.path.exists(probe):
        r = subprocess.run([probe], capture_output=True, text=True, timeout=300)
        print(r.stdout.strip())
        if r.returncode != 0:
            print("# probe failed:", r.stderr[-500:])
    else:
        print("# tools/_ab/mfma_ceiling_probe not built")
    dev = torch.device("cuda:0")
    for dt in (torch.float16, torch.bfloat16):
        for n in (8192, 4096):
            a = torch.randn(n, n, device=dev).to(dt)
            w = (torch.randn(n, n, device=dev) * n ** -0.5).to(dt)
            res = torch.randn(n, n, device=dev)
            out_h = torch.empty(n, n, device=dev, dtype=dt)
            out_f = torch.empty(n, n, device=dev)
            fl = 2 * n ** 3
            t = timed(lambda: ops.gemm(a, w, out=out_h))
            print(f"gemm {n}^3 {str(dt)[6:]:9s} half out                 {t*1e3:7.3f} ms  {fl/t/1e12:7.1f} TFLOP/s")
            t = timed(lambda: ops.gemm(a, w, residual=res, out=out_f))
            print(f"gemm {n}^3 {str(dt)[6:]:9s} fp32 out + fp32 residual {t*1e3:7.3f} ms  {fl/t/1e12:7.1f} TFLOP/s")
            az = torch.zeros_like(a)
            t = timed(lambda: ops.gemm(az, w, out=out_h))
            print(f"gemm {n}^3 {str(dt)[6:]:9s} half out, A = zeros      {t*1e3:7.3f} ms  {fl/t/1e12:7.1f} TFLOP/s   (clock give-back, not a rate to price against)")


if __name__ == "__main__":
    main()
