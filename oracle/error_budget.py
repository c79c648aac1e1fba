"""ORACLE — TEST INFRASTRUCTURE ONLY.  Where does the product's distance from the fp32 reference come from?

The product's precision policy stores fp16 exactly where a tensor is an MFMA operand and keeps fp32
everywhere else.  This script applies that policy to the fp32 ORACLE one tensor class at a time (fake quantisation:
t.half().float() on the class's tensors, everything else fp32) and runs BASELINE configs[0] (256x256, 8 frames, 4 DDIM
steps, CFG 3.5, full-size seeded models) — the fixture whose last-step latents sit at 1.13e-3 on the GPU.  The rel-L2 of
each variant's latents against the pure-fp32 run is that class's share of the error budget; 'all' applies every class at
once and should land near the product's measured figure.

    python -m oracle.error_budget [steps] > profiles/r3_error_budget_config1.txt        (about 10 min on 8 cores)

Classes:  W weights of every Linear / conv of both UNets | CONV inputs of the 3x3 convolutions (GroupNorm+SiLU outputs)
          LN inputs of to_q/to_k/to_v, FF1 and proj_in (LayerNorm / GroupNorm outputs) | QKV outputs of to_q/to_k/to_v
          ATT inputs of to_out (attention outputs) | H inputs of FF2 (GEGLU outputs) | Z inputs of proj_out (FF outputs)
          P softmax probabilities before P.V"""
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import models as OM
from . import primitives as OP
from . import synth
from .pipeline import denoise_clip


def q16(t):
    return t.half().float()


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def hooks_for(cls, nets):
    hs = []
    pre = lambda m, args: (q16(args[0]),) + tuple(args[1:])
    post = lambda m, args, out: q16(out)
    for net in nets:
        for name, m in net.named_modules():
            leaf = name.split(".")[-1]
            parent = ".".join(name.split(".")[-3:])
            if not isinstance(m, (nn.Linear, nn.Conv2d)):
                continue
            is3x3 = isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3)
            if cls == "CONV" and is3x3:
                hs.append(m.register_forward_pre_hook(pre))
            elif cls == "LN" and (leaf in ("to_q", "to_k", "to_v", "proj_in") or parent.endswith("net.0.proj")):
                hs.append(m.register_forward_pre_hook(pre))
            elif cls == "QKV" and leaf in ("to_q", "to_k", "to_v"):
                hs.append(m.register_forward_hook(post))
            elif cls == "ATT" and parent.endswith("to_out.0"):
                hs.append(m.register_forward_pre_hook(pre))
            elif cls == "H" and parent.endswith("net.2"):
                hs.append(m.register_forward_pre_hook(pre))
            elif cls == "Z" and leaf == "proj_out":
                hs.append(m.register_forward_pre_hook(pre))
    return hs


class RoundedWeights:
    def __init__(self, nets):
        self.nets = nets

    def __enter__(self):
        self.saved = []
        for net in self.nets:
            for m in net.modules():
                if isinstance(m, (nn.Linear, nn.Conv2d)):
                    self.saved.append((m, m.weight.data))
                    m.weight.data = q16(m.weight.data)

    def __exit__(self, *a):
        for m, w in self.saved:
            m.weight.data = w


_sdpa = F.scaled_dot_product_attention


def sdpa_rounded_p(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False):
    out = torch.empty_like(q)
    for i in range(q.shape[0]):  # one batch row at a time: bounds the score matrix
        s = (q[i] @ k[i].transpose(-1, -2)) * q.shape[-1] ** -0.5
        out[i] = q16(torch.softmax(s, dim=-1)) @ v[i]
    return out


def main(steps=4):
    torch.set_num_threads(8)
    t0 = time.time()
    o3 = synth.build(OM.UNet3DConditionModel, 1234, motion_heads=8, **OM.SD15_UNET_CONFIG)
    o2 = synth.build(OM.UNet2DConditionModel, 1235, **OM.SD15_UNET_CONFIG)
    print(f"# full-size oracle UNets built in {time.time()-t0:.0f} s; BASELINE configs[0]: 256x256, 8 frames, {steps} DDIM steps, CFG 3.5", flush=True)
    g = torch.Generator().manual_seed(11)
    h, Fr = 32, 8
    ehs = torch.randn(1, 1, 768, generator=g)
    ref_lat = torch.randn(1, 4, h, h, generator=g) * 0.8
    bk = torch.randn(1, 4, Fr, h, h, generator=g) * 0.8
    pose = torch.randn(1, 320, Fr, h, h, generator=g) * 0.5
    lat = torch.randn(1, 4, Fr, h, h, generator=g)

    def run():
        sched = OP.DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS)
        return denoise_clip(o2, o3, sched, ehs, ref_lat, bk, pose, lat.clone(), steps, 3.5, return_trajectory=True)[1]

    base = run()
    print(f"# fp32 baseline done at {time.time()-t0:.0f} s", flush=True)
    print(f"{'class':8s} " + " ".join(f"{'step ' + str(i):>10s}" for i in range(steps)))
    nets = (o3, o2)
    total_sq = [0.0] * steps
    for cls in ("W", "CONV", "LN", "QKV", "ATT", "H", "Z", "P", "all"):
        hs, ctx = [], None
        classes = ("CONV", "LN", "QKV", "ATT", "H", "Z") if cls == "all" else (cls,)
        for c in classes:
            hs += hooks_for(c, nets)
        if cls in ("W", "all"):
            ctx = RoundedWeights(nets)
            ctx.__enter__()
        if cls in ("P", "all"):
            OP.F.scaled_dot_product_attention = sdpa_rounded_p
        try:
            traj = run()
        finally:
            OP.F.scaled_dot_product_attention = _sdpa
            for h_ in hs:
                h_.remove()
            if ctx is not None:
                ctx.__exit__()
        errs = [rel_l2(a, b) for a, b in zip(traj, base)]
        if cls != "all":
            total_sq = [t + e * e for t, e in zip(total_sq, errs)]
        print(f"{cls:8s} " + " ".join(f"{e:10.2e}" for e in errs), flush=True)
    print(f"{'rss':8s} " + " ".join(f"{t ** 0.5:10.2e}" for t in total_sq) + "   (root of the summed squares of the single classes)")
    print(f"# done in {time.time()-t0:.0f} s")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
