"""Where do the per-frame stages spend their time?  VAE decode / encode of `--frames` 512x512 frames (and the pose guider) with
every MFMA launch bracketed: per-shape table against max(FLOPs / 1.3 PFLOP/s, bytes / 4.4 TB/s), the rest (GroupNorm, layout)
as the remainder.
  python tools/vae_bound.py [--frames 8] [--size 512]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

MFMA_RATE, HBM_RATE = 1.3e15, 4.4e12


def table(name, fn, iters=3):
    from mimo_amd import ops
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters
    ops.EVENTS, ops.TAGS = [], []
    fn()
    torch.cuda.synchronize()
    ev, tags = ops.EVENTS, ops.TAGS
    ops.EVENTS = ops.TAGS = None
    shapes = {}
    for (n, a, b, fl, nb), tag in zip(ev, tags):
        r = shapes.setdefault((n, tag), [0, 0.0, 0.0, 0.0, 0.0])
        r[0] += 1
        r[1] += a.elapsed_time(b)
        r[2] += max(fl / MFMA_RATE, nb / HBM_RATE) * 1e3
        r[3] += fl
        r[4] += nb
    tm = sum(r[1] for r in shapes.values())
    tb = sum(r[2] for r in shapes.values())
    print(f"== {name}: {t:.2f} ms; MFMA launches {tm:.2f} ms (their ceilings {tb:.2f} ms), everything else {t - tm:.2f} ms")
    print(f"{'shape':70s} {'n':>3s} {'ms':>7s} {'bound':>7s} {'excess':>7s} {'TFLOP/s':>8s} {'TB/s':>6s}")
    for (n, tag), (cnt, ms, bd, fl, nb) in sorted(shapes.items(), key=lambda kv: kv[1][2] - kv[1][1]):
        print(f"{(n + ' ' + tag)[:70]:70s} {cnt:3d} {ms:7.2f} {bd:7.2f} {ms - bd:7.2f} {fl / ms / 1e9:8.0f} {nb / ms / 1e9:6.2f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--size", type=int, default=512)
    a = ap.parse_args()
    from mimo_amd.vae import AutoencoderKL, PoseGuider
    dev, dt = torch.device("cuda:0"), torch.float16
    with torch.device(dev):
        vae, pg = AutoencoderKL(), PoseGuider()
    for m in (vae, pg):
        m.to(dtype=dt)
        m.compute_dtype = dt
    h = a.size // 8
    z = torch.randn(a.frames, h, h, 8, device=dev).to(dt)
    z[..., 4:] = 0
    img = torch.rand(a.frames, a.size, a.size, 8, device=dev).to(dt)
    img[..., 3:] = 0
    table(f"VAE decode, {a.frames} frames {a.size}x{a.size}", lambda: vae.decode_tokens(z))
    table(f"VAE encode, {a.frames} frames {a.size}x{a.size}", lambda: vae.encode_tokens(img))
    if hasattr(pg, "run_tokens"):
        table(f"pose guider, {a.frames} frames", lambda: pg.run_tokens(img))


if __name__ == "__main__":
    main()
