"""BASELINE configs[0] golden fixture under variants of the default policy: ops.EDGE_SPLIT bits (1 output head, 2 per-clip tables,
4 input convolution), the reference UNet / pose guider under the split policy.  GPU box:  python tools/config1_probe.py"""
import os
import sys

import torch
from safetensors.torch import load_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mimo_amd import ops  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    from mimo_amd.pipeline import Pose2VideoPipeline
    from mimo_amd.scheduler import DDIMScheduler
    from mimo_amd.unet import UNet2DConditionModel, UNet3DConditionModel
    from mimo_amd.vae import AutoencoderKL, PoseGuider
    from oracle import models as OM, primitives as OP, synth
    torch.set_num_threads(min(os.cpu_count(), 32))
    m = {}
    for name, ocls, pcls, seed, okw in (("den", OM.UNet3DConditionModel, UNet3DConditionModel, 1234, dict(motion_heads=8)),
                                        ("ref", OM.UNet2DConditionModel, UNet2DConditionModel, 1235, {}),
                                        ("pose", OM.PoseGuider, PoseGuider, 1236, {}), ("vae", OP.AutoencoderKL, AutoencoderKL, 1237, {})):
        o = synth.build(ocls, seed, **okw)
        p_ = pcls()
        p_.load_state_dict(o.state_dict(), strict=True)
        del o
        p_.to(torch.device("cuda:0"))
        p_.compute_dtype = torch.float16
        m[name] = p_
    G = load_file(os.path.join(ROOT, "tests", "golden", "config1_256_8f_4steps.safetensors"))
    dev = torch.device("cuda:0")
    H = W = 256
    F = 8
    g = torch.Generator().manual_seed(11)
    ref_img = torch.rand(1, 3, H, W, generator=g) * 2 - 1
    bk = torch.ones(F, 3, H, W)
    pose = torch.rand(F, 3, H, W, generator=g)
    clip = torch.randn(1, 768, generator=g)
    lat = torch.randn(1, 4, F, H // 8, W // 8, generator=g)
    for name, knob, refp, posep in (("EDGE_SPLIT = 0", 0, "half", "half"), ("EDGE_SPLIT = 7 (head, tables, conv_in)", 7, "half", "half"),
                                    ("EDGE_SPLIT = 15 (+ last resnet's shortcut)", 15, "half", "half"),
                                    ("EDGE_SPLIT = 15 + reference UNet split", 15, "split", "half")):
        ops.EDGE_SPLIT = knob
        m["ref"].precision = refp
        m["pose"].precision = posep
        pipe = Pose2VideoPipeline(m["vae"], None, m["ref"], m["den"], m["pose"], DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS))
        traj = []
        pipe.run_tensors(ref_img.to(dev), bk.to(dev), pose.to(dev), clip.to(dev), lat.to(dev), 4, 3.5, trajectory=traj, decode=False)
        print(f"{name:44s} " + " ".join("%.3e" % rel(traj[i].cpu(), G[f"latents_step{i}"]) for i in range(4)), flush=True)
    m["ref"].precision = "half"


if __name__ == "__main__":
    main()
