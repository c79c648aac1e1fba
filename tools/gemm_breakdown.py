"""Per-shape breakdown of the gemm_kernel family inside ONE denoising forward (CFG batch of 2 x 24 latent frames):
every ops.gemm / ops.conv2d call is bracketed by HIP events on the launch stream and grouped by its shape and epilogue.
  python tools/gemm_breakdown.py [--size 512] [--bf16]
Prints, per shape: calls, total ms, average us, TF/s, share of the family's time."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mimo_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--bf16", action="store_true")
    a = ap.parse_args()
    dtype = torch.bfloat16 if a.bf16 else torch.float16
    dev = torch.device("cuda:0")
    pipe = bench.build_pipeline(dev, dtype)
    rec = []
    g0, c0 = ops.gemm, ops.conv2d

    def gemm(x, w, **kw):
        M, K = x.shape
        N = w.shape[0]
        tag = f"gemm M{M} N{N} K{K}" + "".join(f" {k}" for k in ("geglu", "silu", "out_f32") if kw.get(k)) + \
              (" res" if kw.get("residual") is not None else "") + (" ln" if kw.get("ln") else "") + \
              (" cs" if kw.get("colstats") else "") + (" imgb" if kw.get("img_bias") is not None else "")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = g0(x, w, **kw)
        e1.record()
        rec.append((tag, e0, e1, 2 * M * N * K * (1 if not kw.get("geglu") else 1)))
        return r

    def conv2d(x, w, cout, **kw):
        n, H, W, cin = x.shape
        ks, st = kw.get("ksize", 3), kw.get("stride", 1)
        up = kw.get("upsample_to")
        x2 = kw.get("x2")
        tag = f"conv{ks}x{ks} n{n} {H}x{W} {cin}->{cout}" + (f" s{st}" if st != 1 else "") + (f" up{up[0]}" if up else "") + \
              (f" +sc{x2.shape[-1]}" if x2 is not None else "") + (" cs" if kw.get("colstats") else "") + \
              (" res" if kw.get("residual") is not None else "") + (" out_f32" if kw.get("out_f32") else "")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = c0(x, w, cout, **kw)
        e1.record()
        Ho, Wo = r.shape[1], r.shape[2]
        rec.append((tag, e0, e1, 2 * n * Ho * Wo * cout * (ks * ks * cin + (x2.shape[-1] if x2 is not None else 0))))
        return r

    # warm-up + banks through the ordinary path, then one recorded forward
    bench.measure_forward(pipe, dev, dtype, a.size, iters=1)
    ops.gemm, ops.conv2d = gemm, conv2d
    try:
        t, fl, n, fam = bench.measure_forward(pipe, dev, dtype, a.size, iters=1)
    finally:
        ops.gemm, ops.conv2d = g0, c0
    torch.cuda.synchronize()
    # measure_forward runs the forward several times (count, timed, events): keep the LAST forward's records
    per = len(rec) // 3 if len(rec) % 3 == 0 else len(rec)
    rec = rec[-per:]
    agg = {}
    for tag, e0, e1, f in rec:
        d = agg.setdefault(tag, [0, 0.0, 0])
        d[0] += 1
        d[1] += e0.elapsed_time(e1)
        d[2] += f
    tot = sum(d[1] for d in agg.values())
    print(f"# size {a.size}: {len(rec)} gemm/conv calls in one forward, {tot:.2f} ms inside their event brackets")
    print(f"{'shape':58s} {'calls':>5s} {'ms':>8s} {'avg us':>8s} {'TF/s':>7s} {'share':>6s}")
    for tag, d in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{tag:58s} {d[0]:5d} {d[1]:8.3f} {d[1] * 1e3 / d[0]:8.1f} {d[2] / d[1] / 1e9:7.0f} {d[1] / tot * 100:5.1f}%")


if __name__ == "__main__":
    main()
