"""Structured probes of mimo_conv3x3_fused (GPU box): one-hot weights isolate the tap / channel / pixel mapping.
    python tools/hconv_debug.py"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mimo_amd import ops  # noqa: E402
from mimo_amd.packing import pack_conv  # noqa: E402


def ref(x, w, dt, ups=False):
    xt = x.to(dt).float().permute(0, 3, 1, 2)
    if ups:
        xt = F.interpolate(xt, scale_factor=2, mode="nearest")
    return F.conv2d(xt, w.float(), padding=1).permute(0, 2, 3, 1)


def report(tag, out, r):
    d = (out - r).abs()
    rel = float((out - r).norm() / r.norm().clamp_min(1e-30))
    print(f"{tag}: rel {rel:.2e}  max|d| {float(d.max()):.3e}")
    if rel > 1e-3:
        bad = d > 1e-2 * r.abs().max()
        n, H, W, N = out.shape
        print("   bad fraction", float(bad.float().mean()), " by row", [round(float(bad[0, y].float().mean()), 2) for y in range(min(H, 18))])
        print("   by col", [round(float(bad[0, :, x].float().mean()), 2) for x in range(min(W, 18))])
        print("   by channel (first 32)", [round(float(bad[0, :, :, c].float().mean()), 2) for c in range(min(N, 32))])
        print("   out[0,0,0,:8]", out[0, 0, 0, :8].tolist(), "\n   ref[0,0,0,:8]", r[0, 0, 0, :8].tolist())
        print("   out[0,5,7,:8]", out[0, 5, 7, :8].tolist(), "\n   ref[0,5,7,:8]", r[0, 5, 7, :8].tolist())


def main():
    dev = torch.device("cuda:0")
    dt = torch.float16
    torch.manual_seed(0)
    for (C, cout, H, W) in [(32, 128, 16, 16), (64, 128, 16, 16), (128, 128, 32, 32), (64, 320, 16, 16), (192, 256, 32, 16)]:
        x = torch.randn(1, H, W, C, device=dev)
        # 1: identity through the centre tap
        w = torch.zeros(cout, C, 3, 3, device=dev)
        for c in range(min(C, cout)):
            w[c, c, 1, 1] = 1.0
        report(f"C{C} N{cout} {H}x{W} centre identity", ops.conv3x3_fused(x, pack_conv(w, dt), cout), ref(x, w, dt))
        # 2: each tap alone, channel 0 -> output channel = tap index
        w = torch.zeros(cout, C, 3, 3, device=dev)
        for t in range(9):
            w[t, 0, t // 3, t % 3] = 1.0
        report(f"C{C} N{cout} {H}x{W} one tap per output", ops.conv3x3_fused(x, pack_conv(w, dt), cout), ref(x, w, dt))
        # 3: random
        w = torch.randn(cout, C, 3, 3, device=dev) * 0.05
        report(f"C{C} N{cout} {H}x{W} random", ops.conv3x3_fused(x, pack_conv(w, dt), cout), ref(x, w.to(dt), dt))
    x = torch.randn(1, 8, 8, 64, device=dev)
    w = torch.randn(128, 64, 3, 3, device=dev) * 0.05
    report("upsample 8->16", ops.conv3x3_fused(x, pack_conv(w, dt), 128, upsample2x=True), ref(x, w.to(dt), dt, True))


if __name__ == "__main__":
    main()
