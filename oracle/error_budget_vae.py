"""ORACLE — TEST INFRASTRUCTURE ONLY.  Error budget of the VAE (sd-vae-ft-mse topology, full size, seeded weights): the
product's storage policy applied to the fp32 oracle ONE tensor class at a time (fake quantisation t.half().float()),
encoder and decoder separately; rel-L2 of the posterior mean / the decoded image against the pure-fp32 run.

    python -m oracle.error_budget_vae [size] > profiles/r4_error_budget_vae.txt      (a few minutes on 8 cores at 256)

Classes:  W     weights of every conv / Linear (what weight_dtype fp16 does in the reference too)
          IN    the input image / latent of the first convolution
          CONV  inputs of the 3x3 convolutions inside the res blocks and of conv_out (GroupNorm + SiLU outputs)
          SAMP  inputs of the down- / up-sampling convolutions (raw activations, no normalisation in front)
          SC    inputs of the 1x1 shortcut convolutions (raw activations)
          GNA   inputs of to_q / to_k / to_v of the mid-block attention (GroupNorm outputs)
          QKV   outputs of to_q / to_k / to_v;   P  softmax probabilities;   ATT  inputs of to_out
          MOM   the encoder's conv_out output (the 8 moment channels quant_conv consumes) / the decoder's post_quant_conv output"""
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import primitives as OP
from . import synth


def q16(t):
    return t.half().float()


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def hooks_for(cls, net, part):
    hs = []
    pre = lambda m, args: (q16(args[0]),) + tuple(args[1:])
    post = lambda m, args, out: q16(out)
    root = getattr(net, part)
    for name, m in root.named_modules():
        leaf = name.split(".")[-1]
        if not isinstance(m, (nn.Linear, nn.Conv2d)):
            continue
        is3 = isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3)
        sampler = "samplers" in name
        if cls == "IN" and name == "conv_in":
            hs.append(m.register_forward_pre_hook(pre))
        elif cls == "CONV" and is3 and not sampler and name != "conv_in":
            hs.append(m.register_forward_pre_hook(pre))
        elif cls == "SAMP" and sampler:
            hs.append(m.register_forward_pre_hook(pre))
        elif cls == "SC" and leaf == "conv_shortcut":
            hs.append(m.register_forward_pre_hook(pre))
        elif cls == "GNA" and leaf in ("to_q", "to_k", "to_v"):
            hs.append(m.register_forward_pre_hook(pre))
        elif cls == "QKV" and leaf in ("to_q", "to_k", "to_v"):
            hs.append(m.register_forward_hook(post))
        elif cls == "ATT" and name.endswith("to_out.0"):
            hs.append(m.register_forward_pre_hook(pre))
        elif cls == "MOM" and part == "encoder" and name == "conv_out":
            hs.append(m.register_forward_hook(post))
    if cls == "MOM" and part == "decoder":
        hs.append(net.post_quant_conv.register_forward_hook(post))
    return hs


class RoundedWeights:
    def __init__(self, net):
        self.net = net

    def __enter__(self):
        self.saved = []
        for m in self.net.modules():
            if isinstance(m, (nn.Linear, nn.Conv2d)):
                self.saved.append((m, m.weight.data))
                m.weight.data = q16(m.weight.data)

    def __exit__(self, *a):
        for m, w in self.saved:
            m.weight.data = w


_sdpa = F.scaled_dot_product_attention


def sdpa_rounded_p(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False):
    s = (q @ k.transpose(-1, -2)) * q.shape[-1] ** -0.5
    return q16(torch.softmax(s, dim=-1)) @ v


def main(size=256):
    torch.set_num_threads(8)
    t0 = time.time()
    vae = synth.build(OP.AutoencoderKL, 4321)
    g = torch.Generator().manual_seed(3)
    img = torch.rand(1, 3, size, size, generator=g) * 2 - 1
    lat = torch.randn(1, 4, size // 8, size // 8, generator=g)
    print(f"# full-size oracle AutoencoderKL (seeded), {size}x{size}: encode of a uniform random image, decode of a normal latent")
    runs = {"encoder": lambda: vae.encode(img).latent_dist.mean, "decoder": lambda: vae.decode(lat).sample}
    classes = ("W", "IN", "CONV", "SAMP", "SC", "GNA", "QKV", "P", "ATT", "MOM", "all")
    for part, run in runs.items():
        with torch.no_grad():
            base = run()
        tot = 0.0
        print(f"{part}:")
        for cls in classes:
            hs, ctx = [], None
            for c in (classes[1:-1] if cls == "all" else (cls,)):
                hs += hooks_for(c, vae, part)
            if cls in ("W", "all"):
                ctx = RoundedWeights(vae)
                ctx.__enter__()
            if cls in ("P", "all"):
                OP.F.scaled_dot_product_attention = sdpa_rounded_p
            try:
                with torch.no_grad():
                    out = run()
            finally:
                OP.F.scaled_dot_product_attention = _sdpa
                for h_ in hs:
                    h_.remove()
                if ctx is not None:
                    ctx.__exit__()
            e = rel_l2(out, base)
            if cls != "all":
                tot += e * e
            print(f"  {cls:6s} {e:10.2e}", flush=True)
        print(f"  {'rss':6s} {tot ** 0.5:10.2e}   (root of the summed squares of the single classes)")
    print(f"# done in {time.time()-t0:.0f} s")


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:2]))
