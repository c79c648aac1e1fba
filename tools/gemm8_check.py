"""Correctness + timing of the 8-phase dense GEMM (gemm8_kernel) against the shipped tile kernels, in one process through the
tune build (MIMO_GEMM_8P is read at every launch):   python tools/gemm8_check.py"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIMO_HIP_LIB", os.path.join(ROOT, "mimo_amd", "libmimo_hip_tune.so"))
from mimo_amd import ops  # noqa: E402
from mimo_amd.packing import pack_geglu  # noqa: E402


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters * 1e-3


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    dev, dt = torch.device("cuda:0"), torch.float16
    g = torch.Generator().manual_seed(0)
    # correctness on small ragged problems (M, N not multiples of 256), every epilogue the token GEMMs use
    for (M, N, K) in [(256 * 8 * 9 + 77, 1280, 640), (256 * 33, 1920, 1280), (4100, 2560 + 64, 128)]:
        a = torch.randn(M, K, generator=g).to(dev).to(dt)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev).to(dt)
        b = torch.randn(N, generator=g).to(dev)
        r = torch.randn(M, N, generator=g).to(dev)
        ref = a.float() @ w.float().t()
        os.environ["MIMO_GEMM_8P"] = "1"
        o1 = ops.gemm(a, w, bias=b)
        o2 = ops.gemm(a, w, bias=b, residual=r, out_f32=True)
        for _ in range(20):  # a race in the slot hand-off would show as run-to-run differences
            assert torch.equal(ops.gemm(a, w, bias=b), o1) and torch.equal(ops.gemm(a, w, bias=b, residual=r, out_f32=True), o2)
        os.environ["MIMO_GEMM_8P"] = "0"
        p1 = ops.gemm(a, w, bias=b)
        print(f"M{M} N{N} K{K}: half out rel {rel(o1.float(), ref + b):.2e} (tiled {rel(p1.float(), ref + b):.2e}); f32+res rel {rel(o2, ref + b + r):.2e}", flush=True)
        if N % 32 == 0:
            wg = (torch.randn(2 * N, K, generator=g) * K ** -0.5).to(dev).to(dt)
            bg = torch.randn(2 * N, generator=g).to(dev) * 0.1
            wp, bp = pack_geglu(wg, bg, dt)
            h = a.float() @ wg.float().t() + bg
            refg = h[:, :N] * F.gelu(h[:, N:])
            os.environ["MIMO_GEMM_8P"] = "1"
            og = ops.gemm(a, wp, bias=bp, geglu=True)
            print(f"   geglu rel {rel(og.float(), refg):.2e}", flush=True)
    # timing at the forward's shapes, interleaved
    print(f"{'shape':40s} {'tiled ms':>9s} {'TF/s':>7s} {'8-phase ms':>10s} {'TF/s':>7s} {'8p 1 blk/tile':>13s} {'TF/s':>7s}   (8-phase: persistent blocks)")
    for (M, N, K, geglu, res) in [(49152, 5120, 640, True, False), (12288, 10240, 1280, True, False), (49152, 1920, 640, False, False),
                                  (12288, 3840, 1280, False, False), (49152, 640, 2560, False, True), (12288, 1280, 5120, False, True),
                                  (12288, 1280, 1280, False, True), (3072, 10240, 1280, True, False), (3072, 3840, 1280, False, False)]:
        a = torch.randn(M, K, device=dev).to(dt)
        w = (torch.randn(N, K, device=dev) * 0.02).to(dt)
        b = torch.zeros(N, device=dev)
        r = torch.randn(M, N, device=dev) if res else None
        fn = lambda: ops.gemm(a, w, bias=b, geglu=geglu, residual=r, out_f32=res)
        best = [1e9, 1e9, 1e9]
        for _ in range(4):
            for i, v in enumerate(("0", "1", "2")):
                os.environ["MIMO_GEMM_8P"] = v
                best[i] = min(best[i], timeit(fn))
        fl = 2 * M * N * K
        print(f"gemm M{M} N{N} K{K}{' geglu' if geglu else ''}{' +res f32' if res else ''}".ljust(40) +
              f" {best[0]*1e3:9.3f} {fl/best[0]/1e12:7.0f} {best[1]*1e3:10.3f} {fl/best[1]/1e12:7.0f} {best[2]*1e3:13.3f} {fl/best[2]/1e12:7.0f}", flush=True)
    os.environ["MIMO_GEMM_8P"] = "0"


if __name__ == "__main__":
    main()
