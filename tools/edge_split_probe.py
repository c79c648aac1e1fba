"""Does split precision at the EDGES of the UNet (output head, per-clip tables) move the forward's distance from the fp32 oracle?
Half-width test models, one forward with bank, default policy with ops.EDGE_SPLIT = 0 / 1 / 2 / 3.  GPU box."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import build_pair_unets  # noqa: E402
from mimo_amd import ops  # noqa: E402
from mimo_amd.unet import ReferenceAttentionControl  # noqa: E402
from oracle import models as OM  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    dev = torch.device("cuda:0")
    o3, o2, p3, p2 = build_pair_unets(torch.float16, dev)
    g = torch.Generator().manual_seed(6)
    hw, F = 16, 8
    ehs = torch.cat([torch.zeros(1, 1, 768), torch.randn(1, 1, 768, generator=g)])
    ref_lat = torch.randn(1, 4, hw, hw, generator=g)
    x = torch.randn(2, 8, F, hw, hw, generator=g)
    pose = torch.randn(2, 160, F, hw, hw, generator=g)
    with torch.no_grad():
        w = OM.ReferenceAttentionControl(o2, "write")
        r = OM.ReferenceAttentionControl(o3, "read")
        o2(ref_lat.repeat(2, 1, 1, 1), torch.zeros(()), ehs)
        r.update(w)
        ref = o3(x, torch.tensor(749), ehs, pose_cond_fea=pose)
    pw = ReferenceAttentionControl(p2, mode="write", do_classifier_free_guidance=True)
    pr = ReferenceAttentionControl(p3, mode="read", do_classifier_free_guidance=True)
    p2(ref_lat.repeat(2, 1, 1, 1).to(dev), 0, ehs.to(dev), stop_after=pw.last_block())
    pr.update(pw)
    for knob in (0, 1, 4, 5):
        ops.EDGE_SPLIT = knob
        out = p3(x.to(dev), 749, ehs.to(dev), pose_cond_fea=pose.to(dev), return_dict=False)[0].float().cpu()
        gd = out[0] + 3.5 * (out[1] - out[0])
        gr = ref[0] + 3.5 * (ref[1] - ref[0])
        print(f"EDGE_SPLIT={knob}: forward rel-L2 {rel(out, ref):.3e}   guided (u + 3.5 (c - u)) rel-L2 {rel(gd, gr):.3e}", flush=True)


if __name__ == "__main__":
    main()
