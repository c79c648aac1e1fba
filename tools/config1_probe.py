"""BASELINE configs[0] golden fixture under variants of the default policy: ops.EDGE_SPLIT bits (1 output head, 2 per-clip tables,
4 input convolution, 8 conv2 + shortcut of the last two resnets), the reference UNet / pose guider under the split policy;
--scan [--parts]: the final latents with one block group / module / resnet product at a time on split operands.  GPU box:  python tools/config1_probe.py"""
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
    if "--scan" in sys.argv:   # which part of the denoising UNet carries configs[0]'s distance from fp32: one group at a time on split operands
        from mimo_amd.modules import MotionModule, ResnetBlock, SpatialTransformer
        import time
        ops.EDGE_SPLIT = 7   # (the scan's baseline: without bit 3, which is what the scan is about)
        mods = {n: mod for n, mod in m["den"].named_modules() if isinstance(mod, (MotionModule, ResnetBlock, SpatialTransformer))}

        def run(names):
            for n in names:
                mods[n].precision = "split"
            pipe = Pose2VideoPipeline(m["vae"], None, m["ref"], m["den"], m["pose"], DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS))
            traj = []
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pipe.run_tensors(ref_img.to(dev), bk.to(dev), pose.to(dev), clip.to(dev), lat.to(dev), 4, 3.5, trajectory=traj, decode=False)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3
            for n in names:
                mods[n].precision = "half"
            return [rel(traj[i].cpu(), G[f"latents_step{i}"]) for i in range(4)], dt

        run([])
        e0, t0 = run([])
        print(f"{'default policy (EDGE_SPLIT = %d)' % ops.EDGE_SPLIT:44s} " + " ".join("%.3e" % v for v in e0) + f"   {t0:7.1f} ms", flush=True)
        groups = [f"down_blocks.{i}" for i in range(4)] + ["mid_block"] + [f"up_blocks.{i}" for i in range(4)]
        singles = sorted(n for n in mods if n.startswith("up_blocks.3.") or n.startswith("down_blocks.0."))
        kinds = {"all resnets": ResnetBlock, "all spatial transformers": SpatialTransformer, "all motion modules": MotionModule}
        for label, names in ([] if "--parts" in sys.argv else [(g_, [n for n in mods if n.startswith(g_ + ".")]) for g_ in groups] + [(n, [n]) for n in singles] +
                             [(k, [n for n, mod in mods.items() if isinstance(mod, c)]) for k, c in kinds.items()] +
                             [("up_blocks.3 + down_blocks.0", [n for n in mods if n.startswith("up_blocks.3.") or n.startswith("down_blocks.0.")]),
                              ("up_blocks.2 + up_blocks.3", [n for n in mods if n.startswith("up_blocks.3.") or n.startswith("up_blocks.2.")])]):
            e, t = run(names)
            print(f"{label:44s} " + " ".join("%.3e" % v for v in e) + f"   {t - t0:+7.1f} ms   d(err^2) {e0[3] ** 2 - e[3] ** 2:+.3e}", flush=True)
        # which product of the last resnets (precise.resnet's split_parts)
        for names in (["up_blocks.3.resnets.2"], ["up_blocks.3.resnets.1"], ["up_blocks.3.resnets.2", "up_blocks.3.resnets.1"],
                      ["up_blocks.3.resnets.2", "up_blocks.3.resnets.1", "up_blocks.3.resnets.0"]):
            for parts in (("sc",), ("conv2",), ("conv1",), ("sc", "conv2"), ("conv1", "conv2")):
                for n in names:
                    mods[n].split_parts = parts
                e, t = run(names)
                for n in names:
                    del mods[n].split_parts
                print(f"{' + '.join(x[12:] for x in names):30s} {'+'.join(parts):12s} " + " ".join("%.3e" % v for v in e) +
                      f"   d(err^2) {e0[3] ** 2 - e[3] ** 2:+.3e}", flush=True)
        for names in (["up_blocks.3.resnets.2", "up_blocks.3.resnets.1"],):   # weights-only split (operands' low parts zeroed)
            for parts in (("sc", "conv2"), ("conv2",)):
                for n in names:
                    mods[n].split_parts, mods[n].split_probe = parts, "w_only"
                e, t = run(names)
                for n in names:
                    del mods[n].split_parts, mods[n].split_probe
                print(f"{' + '.join(x[12:] for x in names):30s} {'+'.join(parts):12s} WEIGHTS ONLY " + " ".join("%.3e" % v for v in e), flush=True)
        for names in (["up_blocks.3.resnets.2", "up_blocks.3.attentions.2"], ["up_blocks.3.resnets.2", "up_blocks.3.resnets.1", "up_blocks.3.attentions.2"],
                      ["up_blocks.3.resnets.2", "up_blocks.3.attentions.2", "up_blocks.3.motion_modules.2"]):
            e, t = run(names)
            print(f"{' + '.join(x[12:] for x in names):44s} " + " ".join("%.3e" % v for v in e) + f"   d(err^2) {e0[3] ** 2 - e[3] ** 2:+.3e}", flush=True)
        return
    for name, knob, refp, posep in (("EDGE_SPLIT = 0", 0, "half", "half"), ("EDGE_SPLIT = 7 (head, tables, conv_in)", 7, "half", "half"),
                                    ("EDGE_SPLIT = 15 (+ conv2 / shortcut of the last two resnets)", 15, "half", "half"),
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
