"""Why do the token GEMMs of levels 2-3 run 15-50 % slower inside a forward than in the isolated microbenchmark?
For each shape: the launch's duration (HIP events around the launch only)
  in situ      median over the launches of that shape inside one denoising forward (mimo_amd.ops.EVENTS / TAGS)
  warm         isolated, the same A / residual / output / weight buffers every iteration (the microbenchmark: everything
               the launch touches is still in L2 / the 256 MB Infinity Cache from the previous iteration)
  cold act     isolated, the activation-side buffers (A, residual, output) rotate through a pool larger than the Infinity
               Cache, the weight stays the same
  cold all     activations AND weights rotate (what a forward sees: every buffer was last touched many launches ago)
  produced     cold all, but A is written by a kernel right before the launch (as the LayerNorm / attention in front of the
               GEMM does in a forward): A comes from the producer's L2 / the Infinity Cache, everything else from HBM
    python tools/insitu_gap.py > profiles/r4_insitu_gap.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mimo_amd import ops  # noqa: E402


def med(ts):
    ts = sorted(ts)
    return ts[len(ts) // 2]


def timed(call, iters=30, pre=None):
    ev = []
    for i in range(iters + 4):
        if pre is not None:
            pre(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call(i)
        e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    return med([a.elapsed_time(b) for a, b in ev[4:]]) * 1e3


def insitu_times(size=512):
    from mimo_amd.modules import Ctx, EarlyExit
    from mimo_amd.unet import ReferenceAttentionControl
    dev, dtype = torch.device("cuda:0"), torch.float16
    pipe = bench.build_pipeline(dev, dtype)
    bench.measure_forward(pipe, dev, dtype, size, iters=2)
    h = size // 8
    unet, refu = pipe.denoising_unet, pipe.reference_unet
    g = torch.Generator(device="cpu").manual_seed(7)
    writer = ReferenceAttentionControl(refu, mode="write", do_classifier_free_guidance=True)
    reader = ReferenceAttentionControl(unet, mode="read", do_classifier_free_guidance=True)
    ehs = torch.cat([torch.zeros(1, 1, 768), torch.randn(1, 1, 768, generator=g)]).to(dev)
    rctx = Ctx(dtype, 1, 1)
    rctx.stop_after = writer.last_block()
    try:
        refu.run_tokens(torch.randn(1, h, h, 8, generator=g).to(dev).to(dtype), 0, ehs[1:], 1, 1, None, rctx)
    except EarlyExit:
        pass
    reader.update(writer)
    x = torch.randn(48, h, h, 8, generator=g).to(dev).to(dtype)
    pose = torch.randn(48, h, h, 320, generator=g).to(dev)
    per = {}
    for _ in range(3):
        ops.EVENTS, ops.TAGS = [], []
        unet.run_tokens(x, 499, ehs, 2, 24, pose)
        torch.cuda.synchronize()
        ev, ops.EVENTS = ops.EVENTS, None
        tags, ops.TAGS = ops.TAGS, None
        for (n, e0, e1, fl, nb), tag in zip(ev, tags):
            per.setdefault(tag, []).append(e0.elapsed_time(e1) * 1e3)
    reader.clear()
    writer.clear()
    del pipe
    torch.cuda.empty_cache()
    return {k: (med(v), len(v) // 3) for k, v in per.items()}


def main():
    dev, dt = torch.device("cuda:0"), torch.float16
    situ = insitu_times()
    print(f"{'shape':34s} {'n/fwd':>5s} {'in situ':>8s} {'warm':>8s} {'cold act':>9s} {'cold all':>9s} {'produced':>9s}   us;  "
          f"share of the (in situ - warm) gap explained by cold all / produced")
    for (M, N, K, res, tag) in [(3072, 1280, 1280, True, "gemm M3072 N1280 K1280 f32"), (12288, 1280, 1280, True, "gemm M12288 N1280 K1280 f32"),
                                (3072, 3840, 1280, False, "gemm M3072 N3840 K1280"), (12288, 3840, 1280, False, "gemm M12288 N3840 K1280"),
                                (3072, 1280, 5120, True, "gemm M3072 N1280 K5120 f32"), (12288, 1280, 5120, True, "gemm M12288 N1280 K5120 f32"),
                                (49152, 640, 640, True, "gemm M49152 N640 K640 f32")]:
        per_set = M * K * 2 + (2 if res else 1) * M * N * (4 if res else 2)
        n_sets = max(3, int(800e6 // per_set) + 1)
        n_w = max(3, int(800e6 // (N * K * 2)) + 1)
        A = [torch.randn(M, K, device=dev).to(dt) for _ in range(n_sets)]
        R = [torch.randn(M, N, device=dev) for _ in range(n_sets)] if res else [None] * n_sets
        O = [torch.empty(M, N, device=dev, dtype=torch.float32 if res else dt) for _ in range(n_sets)]
        W = [(torch.randn(N, K, device=dev) * 0.02).to(dt) for _ in range(n_w)]
        src = torch.randn(M, K, device=dev)
        gam, bet = torch.ones(K, device=dev), torch.zeros(K, device=dev)
        run = lambda a, r, o, w: ops.gemm(a, w, residual=r, out_f32=res, out=o)
        at = lambda i: (A[i % n_sets], R[i % n_sets], O[i % n_sets])
        warm = timed(lambda i: run(A[0], R[0], O[0], W[0]))
        cold_act = timed(lambda i: run(*at(i), W[0]))
        cold_all = timed(lambda i: run(*at(i), W[i % n_w]))

        def produce(i):  # a kernel in front of the GEMM writes A (untimed), as the LayerNorm / attention does in a forward
            A[i % n_sets].copy_(src)  # (a fp32 -> half converting copy: any kernel that writes A)
        prod = timed(lambda i: run(*at(i), W[i % n_w]), pre=produce)
        s, cnt = situ.get(tag, (float("nan"), 0))
        gap = s - warm
        frac = lambda v: f"{100 * (v - warm) / gap:5.0f} %" if gap > 0 else "   n/a"
        print(f"{tag:34s} {cnt:5d} {s:8.1f} {warm:8.1f} {cold_act:9.1f} {cold_all:9.1f} {prod:9.1f}   {frac(cold_all)} / {frac(prod)}", flush=True)
        del A, R, O, W


if __name__ == "__main__":
    main()
