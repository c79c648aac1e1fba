"""Where do the 1.0-1.6e-3 latents errors of the small-model pipeline tests (tests/test_models_gpu.py: F = 1 with CFG, F = 3 at
104 x 104, two wrapped windows) come from?  The same run three ways (GPU box):
  full        the product pipeline as the tests run it
  exact VAE   the product UNets / pose guider, but fed the ORACLE's VAE-encoded reference and background latents
  exact in    ... and the oracle's pose-guider features as well (only the reference UNet + denoising loop are the product's)
    python tools/edge_case_bisect.py > profiles/r4_edge_case_bisect.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import rel_l2  # noqa: E402
from test_models_gpu import build_pair_pose, build_pair_unets, build_pair_vae  # noqa: E402


def main():
    from mimo_amd.pipeline import Pose2VideoPipeline
    from mimo_amd.scheduler import DDIMScheduler
    from oracle import primitives as OP, synth
    from oracle.pipeline import encode_image, run_clip
    dev, dtype = torch.device("cuda:0"), torch.float16
    print(f"{'case':34s} {'full':>9s} {'exact VAE':>10s} {'exact in':>9s}   latents rel_l2 vs the fp32 oracle chain (2 DDIM steps)")
    for (F, guidance, hw, seeds) in [(1, 3.5, 64, (81, 82, 83, 9)), (3, 3.5, 104, (81, 82, 83, 9)), (26, 3.5, 64, (61, 62, 63, 7)), (5, 1.0, 64, (81, 82, 83, 9))]:
        o3, o2, p3, p2 = build_pair_unets(dtype, dev, seed=seeds[0])
        ov, pv = build_pair_vae(dtype, dev, seed=seeds[1])
        og, pg = build_pair_pose(dtype, dev, seed=seeds[2])
        g = torch.Generator().manual_seed(seeds[3])
        H = W = hw
        ref_img = torch.rand(1, 3, H, W, generator=g) * 2 - 1
        bk = torch.rand(F, 3, H, W, generator=g) * 2 - 1
        pose = torch.rand(F, 3, H, W, generator=g)
        clip = torch.randn(1, 768, generator=g)
        lat = torch.randn(1, 4, F, H // 8, W // 8, generator=g)
        with torch.no_grad():
            _, lat_o = run_clip(ov, o2, o3, og, OP.DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS), clip, ref_img, bk, pose, lat, 2, guidance)
            enc_o = {"ref": encode_image(ov, ref_img), "bk": torch.cat([encode_image(ov, bk[i:i + 1]) for i in range(F)])}
            pose_o = og(pose.permute(1, 0, 2, 3)[None])[0].permute(1, 2, 3, 0).contiguous()    # [F, h, w, C0]
        errs = []
        for mode in range(3):
            pipe = Pose2VideoPipeline(pv, None, p2, p3, pg, DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS))
            if mode >= 1:
                def enc(images, _o=enc_o, _n=[0]):
                    src = _o["ref"] if images.shape[0] == 1 and _n[0] == 0 else _o["bk"]
                    _n[0] += 1
                    return src.permute(0, 2, 3, 1).contiguous().to(dev).float()
                pipe._encode_frames = enc
            if mode >= 2:
                pipe.pose_guider = type("P", (), {"compute_dtype": dtype, "run_tokens": staticmethod(lambda tok: pose_o[:tok.shape[0]].to(dev).float())})()
                pipe.vae_batch = 1 << 20
            try:
                _, lat_p = pipe.run_tensors(ref_img.to(dev), bk.to(dev), pose.to(dev), clip.to(dev), lat.to(dev), 2, guidance, return_latents=True)
            except Exception as e:  # (a stand-in the pipeline cannot drive: report, do not hide)
                print(f"   mode {mode}: {type(e).__name__}: {e}", flush=True)
                errs.append(float("nan"))
                continue
            errs.append(rel_l2(lat_p.cpu(), lat_o))
        print(f"F={F:2d} guidance={guidance} {hw}x{hw}".ljust(34) + " ".join(f"{e:9.2e}" for e in errs), flush=True)


if __name__ == "__main__":
    main()
