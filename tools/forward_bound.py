"""What do the forward's own kernels' ceilings allow?  One denoising forward (CFG batch of 2 x 24 frames, 512x512 or
--size) with every MFMA launch bracketed (mimo_amd.ops.EVENTS: algorithmic FLOPs and bytes per launch):
    bound = sum over GEMM / conv launches of max(FLOPs / MFMA_RATE, bytes / HBM_RATE)
          + sum over attention launches of FLOPs / ATTN_RATE
          + the measured time of everything else (normalisation / temporal attention / layout kernels: HBM passes already
            within 10-25 % of the practical stream rate)
with MFMA_RATE = 1.3 PFLOP/s (what the long-K convolutions of this code base sustain at the power-limited clock),
HBM_RATE = 4.4 TB/s (a plain read-modify-write on this part, profiles/r2_epilogue_io_probe.txt) and ATTN_RATE = 1.0 PFLOP/s
(d = 40 pads Q.K^T to K = 48 and O^T to 48 rows: 83 % of the 1.2 PFLOP/s the guide quotes for d = 128).
  python tools/forward_bound.py [--size 512] [--mfma-rate 1.3] [--attn-rate 1.0] [--launches]   (--launches: every bracketed launch in issue order)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

MFMA_RATE, HBM_RATE, ATTN_RATE = 1.3e15, 4.4e12, 1.0e15


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--mfma-rate", type=float, default=None, help="PFLOP/s to price the MFMA-bound launches at (default 1.3)")
    ap.add_argument("--attn-rate", type=float, default=None, help="PFLOP/s to price spatial attention at (default 1.0)")
    ap.add_argument("--launches", action="store_true", help="also list every bracketed launch in issue order")
    a = ap.parse_args()
    global MFMA_RATE, ATTN_RATE
    if a.mfma_rate:
        MFMA_RATE = a.mfma_rate * 1e15
    if a.attn_rate:
        ATTN_RATE = a.attn_rate * 1e15
    from mimo_amd import ops
    dev, dtype = torch.device("cuda:0"), torch.float16
    pipe = bench.build_pipeline(dev, dtype)
    t_fwd, flops, launches, fam = bench.measure_forward(pipe, dev, dtype, a.size, iters=3)
    # measure_forward leaves nothing behind: bracket one more forward and keep the per-launch records
    from mimo_amd.modules import Ctx, EarlyExit
    from mimo_amd.unet import ReferenceAttentionControl
    h = a.size // 8
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
    ops.EVENTS, ops.TAGS = [], []
    unet.run_tokens(x, 499, ehs, 2, 24, pose)
    torch.cuda.synchronize()
    ev, ops.EVENTS = ops.EVENTS, None
    tags, ops.TAGS = ops.TAGS, None
    t_gemm = sum(e0.elapsed_time(e1) for n, e0, e1, fl, nb in ev if n == "gemm_kernel") * 1e-3
    t_attn = sum(e0.elapsed_time(e1) for n, e0, e1, fl, nb in ev if n == "attn_kernel") * 1e-3
    b_gemm = sum(max(fl / MFMA_RATE, nb / HBM_RATE) for n, e0, e1, fl, nb in ev if n == "gemm_kernel")
    b_gemm_mfma = sum(fl / MFMA_RATE for n, e0, e1, fl, nb in ev if n == "gemm_kernel")
    hbm_bound = sum(1 for n, e0, e1, fl, nb in ev if n == "gemm_kernel" and nb / HBM_RATE > fl / MFMA_RATE)
    b_attn = sum(fl / ATTN_RATE for n, e0, e1, fl, nb in ev if n == "attn_kernel")
    rest = t_fwd - t_gemm - t_attn
    print(f"forward {t_fwd*1e3:.2f} ms: GEMM / conv launches {t_gemm*1e3:.2f} ms, spatial attention {t_attn*1e3:.2f} ms, everything else {rest*1e3:.2f} ms")
    print(f"GEMM / conv at their own ceilings: {b_gemm*1e3:.2f} ms ({hbm_bound} of {sum(1 for r in ev if r[0] == 'gemm_kernel')} launches HBM-bound at "
          f"{HBM_RATE/1e12:.1f} TB/s; MFMA-only bound at {MFMA_RATE/1e15:.1f} PFLOP/s: {b_gemm_mfma*1e3:.2f} ms)")
    print(f"spatial attention at {ATTN_RATE/1e15:.1f} PFLOP/s: {b_attn*1e3:.2f} ms")
    print(f"bound of this op decomposition: {(b_gemm + b_attn + rest)*1e3:.2f} ms  (measured {t_fwd*1e3:.2f} ms = {t_fwd/(b_gemm + b_attn + rest):.2f} x)")
    # per-shape table of the GEMM / conv launches, by time above their own ceiling
    shapes = {}
    for (n, e0, e1, fl, nb), tag in zip(ev, tags):
        if n != "gemm_kernel":
            continue
        r = shapes.setdefault(tag, [0, 0.0, 0.0, 0.0, 0.0])
        r[0] += 1
        r[1] += e0.elapsed_time(e1)
        r[2] += max(fl / MFMA_RATE, nb / HBM_RATE) * 1e3
        r[3] += fl
        r[4] += nb
    print(f"{'shape':64s} {'n':>3s} {'ms':>7s} {'bound':>7s} {'excess':>7s} {'TFLOP/s':>8s} {'TB/s':>6s}")
    for tag, (cnt, ms, bd, fl, nb) in sorted(shapes.items(), key=lambda kv: kv[1][2] - kv[1][1]):
        print(f"{tag:64s} {cnt:3d} {ms:7.2f} {bd:7.2f} {ms - bd:7.2f} {fl / ms / 1e9:8.0f} {nb / ms / 1e9:6.2f}")
    if a.launches:
        print("# launches in issue order (tag, us, bound us, TFLOP/s, TB/s)")
        for (n, e0, e1, fl, nb), tag in zip(ev, tags):
            ms = e0.elapsed_time(e1)
            bd = (fl / (ATTN_RATE if n == "attn_kernel" else MFMA_RATE)) if n == "attn_kernel" else max(fl / MFMA_RATE, nb / HBM_RATE)
            print(f"{tag:64s} {ms*1e3:8.1f} {bd*1e6:8.1f} {fl / ms / 1e9:8.0f} {nb / ms / 1e9:6.2f}")
    reader.clear()
    writer.clear()


if __name__ == "__main__":
    main()
