"""A/B of the two ways to run GroupNorm-apply + SiLU + 3x3 convolution (GPU box):
  old: mimo_group_norm_apply (fp32 -> half pass) + mimo_conv2d (row-tiled implicit GEMM, DMA-staged operands)
  new: mimo_group_norm_affine + mimo_conv3x3_fused (halo-tiled, normalisation in the operand path; csrc/hconv.hip)
at the shapes of the 64 x 64 UNet level and of the VAE, interleaved on one box.  TFLOP/s counts the convolution's MACs only.
    python tools/hconv_bench.py [--dtype fp16|bf16] [--quick]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mimo_amd import ops  # noqa: E402
from mimo_amd.packing import pack_conv  # noqa: E402


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--cold", action="store_true", help="rotate the input through a pool larger than the Infinity Cache (what a forward sees)")
    a = ap.parse_args()
    dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    dev = torch.device("cuda:0")
    # (label, n, H, W, C1, C2, cout, gn, ups, want_raw)
    cases = [
        ("unet L0 conv 320->320", 48, 64, 64, 320, 0, 320, True, False, False),
        ("unet L0 conv1 320+320->320 +raw", 48, 64, 64, 320, 320, 320, True, False, True),
        ("unet L0 conv1 320+640->320 +raw", 48, 64, 64, 320, 640, 320, True, False, True),
        ("unet up 32->64 640->640", 48, 64, 64, 640, 0, 640, False, True, False),
        ("vae 512^2 128->128", 8, 512, 512, 128, 0, 128, True, False, False),
        ("vae 256^2 256->256", 8, 256, 256, 256, 0, 256, True, False, False),
        ("vae 128^2 512->512", 8, 128, 128, 512, 0, 512, True, False, False),
        ("vae 64^2 512->512", 8, 64, 64, 512, 0, 512, True, False, False),
        ("vae up 256->512^2 256->256", 4, 512, 512, 256, 0, 256, False, True, False),
        ("unet L1 conv 640->640 (policy: old)", 48, 32, 32, 640, 0, 640, True, False, False),
    ]
    if a.quick:
        cases = cases[:2] + cases[4:5]
    print(f"{'case':36s} {'old gn':>8s} {'old conv':>9s} {'old sum':>8s} {'new':>8s} {'new/old':>8s} {'old TF/s':>9s} {'new TF/s':>9s}   (ms)")
    for (label, n, H, W, C1, C2, cout, gn, ups, want_raw) in cases:
        C = C1 + C2
        Hs, Ws = (H // 2, W // 2) if ups else (H, W)
        nb = max(1, int(600e6 // (n * Hs * Ws * C * 4)) + 1) if a.cold else 1
        pool1 = [torch.randn(n, Hs, Ws, C1, device=dev) for _ in range(nb)]
        pool2 = [torch.randn(n, Hs, Ws, C2, device=dev) if C2 else None for _ in range(nb)]
        x1, x2 = pool1[0], pool2[0]
        it = [0]

        def rot():
            it[0] += 1
            return pool1[it[0] % nb], pool2[it[0] % nb]
        w = pack_conv(torch.randn(cout, C, 3, 3, device=dev) * 0.02, dt)
        b = torch.zeros(cout, device=dev)
        gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        stats = ops.group_norm_stats(x1, groups=32, x2=x2, dtype=dt) if gn else None
        fl = 2 * n * H * W * cout * 9 * C

        def old_gn():
            x1, x2 = rot()
            if gn:
                return ops.group_norm_apply(x1, stats, gamma, beta, groups=32, silu=True, x2=x2, dtype=dt, want_raw=want_raw)[0]
            return ops.group_norm_apply(x1, None, None, None, dtype=dt, want_norm=False, want_raw=True)[1]

        xh = old_gn()

        def old_conv():
            return ops.conv2d(xh, w, cout, bias=b, out_f32=True, upsample_to=(H, W) if ups else None)

        def old():
            h = old_gn()
            return ops.conv2d(h, w, cout, bias=b, out_f32=True, upsample_to=(H, W) if ups else None)

        def new():
            x1, x2 = rot()
            ab = ops.group_norm_affine(stats, gamma, beta, C) if gn else None
            return ops.conv3x3_fused(x1, w, cout, x2=x2, ab=ab, bias=b, upsample2x=ups, want_raw=want_raw)

        o, nw = old(), new()
        nw = nw[0] if want_raw else nw
        err = float((o - nw).norm() / o.norm())
        ts = {"gn": [], "conv": [], "old": [], "new": []}
        for _ in range(3):
            ts["gn"].append(timeit(old_gn))
            ts["conv"].append(timeit(old_conv))
            ts["old"].append(timeit(old))
            ts["new"].append(timeit(new))
        m = {k: min(v) for k, v in ts.items()}
        print(f"{label:36s} {m['gn']*1e3:8.3f} {m['conv']*1e3:9.3f} {m['old']*1e3:8.3f} {m['new']*1e3:8.3f} {m['new']/m['old']:8.3f} "
              f"{fl/m['conv']/1e12:9.1f} {fl/m['new']/1e12:9.1f}   rel diff {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
