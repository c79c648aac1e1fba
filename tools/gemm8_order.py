"""gemm8_kernel tile order: plain row-major (MIMO_G8_GM=1) against groups of G row panels walked row-fastest (=2, 4, 8), tune
library: bit-identity, interleaved timing on COLD operands (rotating through a pool larger than the Infinity Cache, as in situ)
and warm.  GPU box:  python tools/gemm8_order.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIMO_HIP_LIB", os.path.join(ROOT, "mimo_amd", "libmimo_hip_tune.so"))
from mimo_amd import ops  # noqa: E402
from mimo_amd.packing import pack_geglu  # noqa: E402

GMS = ("1", "2", "4", "8")


def timeit(fn, iters, warm=2):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for i in range(iters):
        fn(i)
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters * 1e3


def main():
    dev, dt = torch.device("cuda:0"), torch.float16
    print(f"{'shape':40s} " + " ".join(f"{'GM=' + g + ' us':>10s}" for g in GMS) + "   (cold operands)   | warm: " + " ".join(f"{'GM=' + g:>8s}" for g in GMS))
    for (M, N, K, geglu) in [(12288, 10240, 1280, True), (49152, 5120, 640, True), (12288, 3840, 1280, False), (49152, 1920, 640, False),
                             (12288, 1280, 5120, False), (3072, 10240, 1280, True), (8192, 8192, 8192, False)]:
        pool = max(2, int(600e6 // ((M * K + (M * N // (2 if geglu else 1))) * 2)) + 1) if M * K < 6e7 else 2
        As = [torch.randn(M, K, device=dev).to(dt) for _ in range(pool)]
        if geglu:
            w, b = pack_geglu(torch.randn(N, K, device=dev) * 0.02, torch.zeros(N, device=dev), dt)
        else:
            w, b = (torch.randn(N, K, device=dev) * 0.02).to(dt), torch.zeros(N, device=dev)
        outs = [torch.empty(M, N // 2 if geglu else N, device=dev, dtype=dt) for _ in range(pool)]
        fn = lambda i: ops.gemm(As[i % pool], w, bias=b, geglu=geglu, out=outs[i % pool])
        ref = None
        for gm in GMS:
            os.environ["MIMO_G8_GM"] = gm
            o = ops.gemm(As[0], w, bias=b, geglu=geglu).clone()
            ref = o if ref is None else ref
            assert torch.equal(o, ref), (M, N, K, gm)
        cold, warm = [1e9] * len(GMS), [1e9] * len(GMS)
        for _ in range(4):
            for i, gm in enumerate(GMS):
                os.environ["MIMO_G8_GM"] = gm
                cold[i] = min(cold[i], timeit(fn, 3 * pool))
                warm[i] = min(warm[i], timeit(lambda k: fn(0), 10))
        print(f"gemm M{M} N{N} K{K}{' geglu' if geglu else ''}".ljust(40) + " " + " ".join(f"{t:10.1f}" for t in cold) +
              "                     | " + " ".join(f"{t:8.1f}" for t in warm), flush=True)
    os.environ.pop("MIMO_G8_GM", None)


if __name__ == "__main__":
    main()
