"""Experiment: do two half-batch denoising forwards on two HIP streams overlap better than one batched forward?
One CFG step = uncond half + cond half (24 frames each).  Times, on one GPU:
  (a) the batched b = 2 forward (what the pipeline runs),
  (b) two b = 1 forwards back to back on one stream,
  (c) the same two forwards on two streams concurrently.
(Both halves use the cond configuration here — bank attached — so (b)/(c) do slightly more attention work than (a).)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mimo_amd import ops  # noqa: E402
from mimo_amd.modules import Ctx, EarlyExit  # noqa: E402
from mimo_amd.unet import ReferenceAttentionControl  # noqa: E402


def main():
    dev, dtype, size = torch.device("cuda:0"), torch.float16, 512
    pipe = bench.build_pipeline(dev, dtype)
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
    xs, poses = (x[:24].contiguous(), x[24:].contiguous()), (pose[:24].contiguous(), pose[24:].contiguous())
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def timed(fn, n=3):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    def batched():
        unet.run_tokens(x, 499, ehs, 2, 24, pose)

    def sequential():
        for i in range(2):
            unet.run_tokens(xs[i], 499, ehs[1:], 1, 24, poses[i])

    def concurrent():
        cur = torch.cuda.current_stream()
        for s in (s1, s2):
            s.wait_stream(cur)
        for i, s in enumerate((s1, s2)):
            with torch.cuda.stream(s):
                unet.run_tokens(xs[i], 499, ehs[1:], 1, 24, poses[i])
        for s in (s1, s2):
            cur.wait_stream(s)

    def two_batched_seq():
        batched()
        batched()

    def two_batched_conc():
        cur = torch.cuda.current_stream()
        for s in (s1, s2):
            s.wait_stream(cur)
        for s in (s1, s2):
            with torch.cuda.stream(s):
                unet.run_tokens(x, 499, ehs, 2, 24, pose)
        for s in (s1, s2):
            cur.wait_stream(s)

    def unit(cond):
        blocks = unet.spatial_blocks()
        saved = [b.bank_kv for b in blocks]
        if not cond:
            for b in blocks:
                b.bank_kv = None
        try:
            unet.run_tokens(xs[0], 499, ehs[1:] if cond else ehs[:1], 1, 24, poses[0])
        finally:
            for b, kv in zip(blocks, saved):
                b.bank_kv = kv

    with ops.split_k(False):
        # the items of pipeline.plan_items: a whole window (b = 2) and its CFG halves as b = 1 forwards (pipeline._run_unit)
        best = [1e9, 1e9, 1e9]
        for _ in range(3):
            for i, fn in enumerate((batched, lambda: unit(True), lambda: unit(False))):
                best[i] = min(best[i], timed(fn))
        def window_plus_half(conc):  # what the busiest rank of plan_items(10 windows, 8 ranks) runs per step
            cur = torch.cuda.current_stream()
            if not conc:
                batched()
                unit(True)
                return
            for s in (s1, s2):
                s.wait_stream(cur)
            with torch.cuda.stream(s1), ops.workspace_slot(0):
                batched()
            with torch.cuda.stream(s2), ops.workspace_slot(1):
                unit(True)
            for s in (s1, s2):
                cur.wait_stream(s)
        wp = [min(timed(lambda: window_plus_half(False)) for _ in range(2)), min(timed(lambda: window_plus_half(True)) for _ in range(2))]
        print(f"busiest rank at 8 GPUs (one window b=2 + one cond half b=1): back to back {wp[0]:.2f} ms | on two streams {wp[1]:.2f} ms", flush=True)
        print(f"unit costs (split-K off, as in the sharded mode): window b=2 {best[0]:.2f} ms | cond half b=1 {best[1]:.2f} ms "
              f"({best[1]/best[0]:.3f}) | uncond half b=1 {best[2]:.2f} ms ({best[2]/best[0]:.3f})", flush=True)
    with ops.split_k(False):
        print(f"two windows (b = 2 each): sequential {timed(two_batched_seq):.2f} ms | on two streams {timed(two_batched_conc):.2f} ms", flush=True)
    with ops.split_k(False):  # the split-K workspace is shared: not safe across concurrent streams
        for rnd in range(2):
            print(f"round {rnd}: batched b=2 {timed(batched):.2f} ms | two b=1 sequential {timed(sequential):.2f} ms | "
                  f"two b=1 on two streams {timed(concurrent):.2f} ms", flush=True)


if __name__ == "__main__":
    main()
