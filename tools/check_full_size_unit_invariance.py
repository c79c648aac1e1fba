"""Full-size check of what the sharded long-clip mode relies on: ONE denoising forward of the CFG batch (b = 2 x 24 frames, 512 x 512,
split-K off) equals, BIT FOR BIT, its two halves run as b = 1 forwards (the cond half with the bank, the uncond half without) —
through every fused kernel of the level-0 path (block head, attention core, block tail, fused convolutions).
    python tools/check_full_size_unit_invariance.py [--size 512]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mimo_amd import ops  # noqa: E402
from mimo_amd.modules import Ctx, EarlyExit  # noqa: E402
from mimo_amd.unet import ReferenceAttentionControl  # noqa: E402


def main():
    size = int(sys.argv[sys.argv.index("--size") + 1]) if "--size" in sys.argv else 512
    dev, dtype = torch.device("cuda:0"), torch.float16
    pipe = bench.build_pipeline(dev, dtype)
    unet, refu = pipe.denoising_unet, pipe.reference_unet
    h = size // 8
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
    F = 24
    x1 = torch.randn(F, h, h, 8, generator=g).to(dev).to(dtype)
    p1 = torch.randn(F, h, h, 320, generator=g).to(dev)
    x, pose = x1.repeat(2, 1, 1, 1), p1.repeat(2, 1, 1, 1)
    with ops.split_k(False):
        full = unet.run_tokens(x, 499, ehs, 2, F, pose).clone()
        cond = unet.run_tokens(x1.clone(), 499, ehs[1:], 1, F, p1.clone()).clone()
        unc = pipe._run_unit(unet, x1.clone(), 499, ehs[:1], F, p1.clone(), cond=False).clone()
    ok_u, ok_c = torch.equal(full[:F], unc), torch.equal(full[F:], cond)
    print(f"{size}x{size}, {F} frames: uncond half (b = 1, no bank) == rows [0, F) of the b = 2 forward: {ok_u}; "
          f"cond half (b = 1, bank) == rows [F, 2F): {ok_c}; max |full| = {float(full.abs().max()):.3f}", flush=True)
    reader.clear()
    writer.clear()
    assert ok_u and ok_c


if __name__ == "__main__":
    main()
