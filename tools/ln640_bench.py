"""N = 640 projection + LayerNorm: gemm_ln640_kernel (LayerNorm in the GEMM's epilogue) against GEMM + mimo_layer_norm, at the level-1
shape of configs[1] (M = 48 x 1024 rows, K = 640, fp32 residual in / out), cold operands.
    python tools/ln640_bench.py > profiles/r5_ln640_bench.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimo_amd import ops  # noqa: E402


def timed(fn, pool, iters=20):
    for i in range(3):
        fn(pool[i % len(pool)])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(pool[i % len(pool)])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev, dt = torch.device("cuda:0"), torch.float16
    for M, N, K in ((49152, 640, 640), (49152, 640, 2560)):
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
        b, g, be = torch.randn(N, device=dev), torch.ones(N, device=dev), torch.zeros(N, device=dev)
        pool = [(torch.randn(M, K, device=dev).to(dt), torch.randn(M, N, device=dev)) for _ in range(6)]   # > 256 MB in rotation
        ln = dict(gamma=g, beta=be, eps=1e-5)
        run = lambda p: ops.gemm(p[0], w, bias=b, residual=p[1], out_f32=True, ln=ln)
        best = {}
        for _ in range(3):
            for name, flag in (("GEMM + LayerNorm launch", False), ("gemm_ln640_kernel", True)):
                ops.LN_OUT_640 = flag
                best[name] = min(best.get(name, 1e9), timed(run, pool))
        ops.LN_OUT_640 = False
        print(f"M{M} N{N} K{K} fp32 residual: " + "; ".join(f"{k} {v*1e3:.1f} us" for k, v in best.items()), flush=True)


if __name__ == "__main__":
    main()
