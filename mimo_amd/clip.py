"""CLIP ViT image encoder on the HIP path — SURVEY.md 8(f) rank 1: the last GPU model inside
`Pose2VideoPipeline.__call__` that the reference runs with stock ops.

Drop-in for `transformers.CLIPVisionModelWithProjection` as the reference uses it
(run_animate.py:92-94 builds it with `from_pretrained(image_encoder_path)`;
src/pipelines/pipeline_pose2vid_long_edit_bkfill_roiclip.py:379-385 calls
`self.image_encoder(clip_image.to(device, dtype=self.image_encoder.dtype)).image_embeds`):
same constructor config fields, same state-dict keys (`vision_model.embeddings.*`,
`vision_model.pre_layrnorm.*`, `vision_model.encoder.layers.N.{self_attn.{q,k,v,out}_proj, layer_norm1, mlp.fc1,
mlp.fc2, layer_norm2}.*`, `vision_model.post_layernorm.*`, `visual_projection.weight`), same outputs.

Arithmetic (transformers models/clip/modeling_clip.py, CLIPVisionTransformer): patch conv (stride = kernel, no
bias) -> [cls | patches] + position embedding -> pre-LN -> N x {LN1, MHA(q scaled by d^-1/2), +res, LN2,
fc1, quick_gelu, fc2, +res} -> post-LN of the cls token -> projection (no bias).  Here: the patch conv is a GEMM over
unfolded patches, q/k/v are one GEMM with the softmax scale (x log2 e) folded into W_q / b_q, quick_gelu(x) =
x.sigmoid(1.702 x) is the GEMM's SiLU epilogue on 1.702-scaled fc1 weights with out_scale 1/1.702, the residual
stream is fp32.  No CPU / PyTorch fallback: every contraction and normalisation is a libmimo_hip.so launch.
"""
import json
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import ops
from .modules import LOG2E, HipModule, _f32

_QG = 1.702  # quick_gelu slope


class _Attn(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.k_proj = nn.Linear(dim, dim)
        self.v_proj = nn.Linear(dim, dim)
        self.q_proj = nn.Linear(dim, dim)
        self.out_proj = nn.Linear(dim, dim)


class _MLP(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.fc1 = nn.Linear(dim, inner)
        self.fc2 = nn.Linear(inner, dim)


class CLIPEncoderLayer(HipModule):
    def __init__(self, dim, inner, heads, eps):
        super().__init__()
        self.self_attn = _Attn(dim)
        self.layer_norm1 = nn.LayerNorm(dim, eps=eps)
        self.mlp = _MLP(dim, inner)
        self.layer_norm2 = nn.LayerNorm(dim, eps=eps)
        self.dim, self.heads, self.eps = dim, heads, eps

    def _pack(self, dt):
        a, m = self.self_attn, self.mlp
        g = (self.dim // self.heads) ** -0.5 * LOG2E  # softmax scale, in log2 units, carried by q
        return dict(
            qkv_w=torch.cat([a.q_proj.weight.detach().float() * g, a.k_proj.weight.detach().float(),
                             a.v_proj.weight.detach().float()], 0).to(dt).contiguous(),
            qkv_b=torch.cat([a.q_proj.bias.detach().float() * g, a.k_proj.bias.detach().float(),
                             a.v_proj.bias.detach().float()], 0).contiguous(),
            o_w=a.out_proj.weight.detach().to(dt).contiguous(), o_b=_f32(a.out_proj.bias),
            fc1_w=(m.fc1.weight.detach().float() * _QG).to(dt).contiguous(), fc1_b=(m.fc1.bias.detach().float() * _QG).contiguous(),
            fc2_w=m.fc2.weight.detach().to(dt).contiguous(), fc2_b=_f32(m.fc2.bias),
            n1w=_f32(self.layer_norm1.weight), n1b=_f32(self.layer_norm1.bias),
            n2w=_f32(self.layer_norm2.weight), n2b=_f32(self.layer_norm2.bias))

    def run(self, dtype, h, B, T):
        """h: fp32 residual stream [B*T, dim] -> fp32 [B*T, dim]."""
        p = self.packed(dtype)
        C = self.dim
        n1 = ops.layer_norm(h, p["n1w"], p["n1b"], eps=self.eps, dtype=dtype)
        qkv = ops.gemm(n1, p["qkv_w"], bias=p["qkv_b"]).view(B, T, 3 * C)
        o = ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], self.heads, q_prescaled=True)
        h = ops.gemm(o.view(-1, C), p["o_w"], bias=p["o_b"], residual=h, out_f32=True)
        n2 = ops.layer_norm(h, p["n2w"], p["n2b"], eps=self.eps, dtype=dtype)
        u = ops.gemm(n2, p["fc1_w"], bias=p["fc1_b"], silu=True, out_scale=1.0 / _QG)  # = quick_gelu(fc1(n2))
        return ops.gemm(u, p["fc2_w"], bias=p["fc2_b"], residual=h, out_f32=True)


class _Embeddings(nn.Module):
    def __init__(self, dim, image_size, patch, channels):
        super().__init__()
        self.class_embedding = nn.Parameter(torch.randn(dim))
        self.patch_embedding = nn.Conv2d(channels, dim, kernel_size=patch, stride=patch, bias=False)
        self.position_embedding = nn.Embedding((image_size // patch) ** 2 + 1, dim)
        self.register_buffer("position_ids", torch.arange((image_size // patch) ** 2 + 1).expand((1, -1)), persistent=False)


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layers = nn.ModuleList([CLIPEncoderLayer(cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads,
                                                      cfg.layer_norm_eps) for _ in range(cfg.num_hidden_layers)])


class CLIPVisionTransformer(HipModule):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = _Embeddings(cfg.hidden_size, cfg.image_size, cfg.patch_size, cfg.num_channels)
        self.pre_layrnorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)  # (sic) the checkpoint's key
        self.encoder = _Encoder(cfg)
        self.post_layernorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)
        self.cfg = cfg

    def _pack(self, dt):
        e = self.embeddings
        w = e.patch_embedding.weight.detach().float().flatten(1)      # [dim, 3*p*p], (c, ky, kx) order
        kp = (w.shape[1] + 7) // 8 * 8                                # MFMA operand rows are 16-byte granular
        wp = torch.zeros((w.shape[0], kp), device=w.device, dtype=torch.float32)
        wp[:, :w.shape[1]] = w
        return dict(patch_w=wp.to(dt).contiguous(), kp=kp,
                    tok0=(e.class_embedding.detach().float() + e.position_embedding.weight.detach().float()[0]).contiguous(),
                    pos=e.position_embedding.weight.detach().float()[1:].contiguous(),
                    prw=_f32(self.pre_layrnorm.weight), prb=_f32(self.pre_layrnorm.bias),
                    pow=_f32(self.post_layernorm.weight), pob=_f32(self.post_layernorm.bias))

    def run(self, dtype, pixel_values):
        """pixel_values [B, 3, S, S] (device) -> (last_hidden_state fp32 [B, T, dim], pooled half [B, dim])."""
        cfg = self.cfg
        p = self.packed(dtype)
        B = pixel_values.shape[0]
        ps, g = cfg.patch_size, cfg.image_size // cfg.patch_size
        dim = cfg.hidden_size
        # unfold the non-overlapping patches (layout plumbing): [B, g*g, 3*ps*ps] in (c, ky, kx) order, K padded
        x = pixel_values.to(dtype).reshape(B, cfg.num_channels, g, ps, g, ps).permute(0, 2, 4, 1, 3, 5).reshape(B * g * g, -1)
        a = torch.zeros((B * g * g, p["kp"]), device=x.device, dtype=dtype)
        a[:, :x.shape[1]] = x
        T = g * g + 1
        emb = torch.empty((B, T, dim), device=x.device, dtype=torch.float32)
        emb[:, 0] = p["tok0"]
        # patch GEMM; the position embedding rides in the epilogue as a per-row bias (rows_per_img = 1)
        pos = p["pos"] if B == 1 else p["pos"].repeat(B, 1)
        emb[:, 1:] = ops.gemm(a, p["patch_w"], img_bias=pos, rows_per_img=1, out_f32=True).view(B, g * g, dim)
        h = ops.layer_norm(emb.view(-1, dim), p["prw"], p["prb"], eps=cfg.layer_norm_eps, dtype=dtype, out_f32=True)
        for layer in self.encoder.layers:
            h = layer.run(dtype, h, B, T)
        h = h.view(B, T, dim)
        pooled = ops.layer_norm(h[:, 0].contiguous(), p["pow"], p["pob"], eps=cfg.layer_norm_eps, dtype=dtype)
        return h, pooled


class CLIPVisionModelWithProjection(HipModule):
    """`CLIPVisionModelWithProjection(config)` / `.from_pretrained(dir)`; `model(pixel_values).image_embeds`."""

    def __init__(self, config=None, **kw):
        super().__init__()
        d = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, num_channels=3,
                 image_size=224, patch_size=14, layer_norm_eps=1e-5, projection_dim=768, hidden_act="quick_gelu")
        if config is not None:
            src = config if isinstance(config, dict) else {k: getattr(config, k) for k in d if hasattr(config, k)}
            d.update({k: v for k, v in src.items() if k in d})
        d.update({k: v for k, v in kw.items() if k in d})
        if d["hidden_act"] != "quick_gelu":
            raise NotImplementedError("only the quick_gelu CLIP vision towers (OpenAI ViT-B/L) are implemented")
        if (d["hidden_size"] // d["num_attention_heads"]) not in (40, 64, 80, 160):
            raise NotImplementedError("attention head size must be one of 40, 64, 80, 160")
        self.config = SimpleNamespace(**d)
        self.vision_model = CLIPVisionTransformer(self.config)
        self.visual_projection = nn.Linear(d["hidden_size"], d["projection_dim"], bias=False)
        self.compute_dtype = None

    @classmethod
    def from_pretrained(cls, path, **_):
        """`path`: directory with config.json (CLIPVisionConfig or a full CLIPConfig with `vision_config`) and
        model.safetensors | pytorch_model.bin — the layout of pretrained_weights/image_encoder (README.md:97-117)."""
        cfg = json.load(open(os.path.join(path, "config.json")))
        cfg = dict(cfg.get("vision_config", {}), **{k: v for k, v in cfg.items() if k != "vision_config"}) \
            if "vision_config" in cfg else cfg
        model = cls(cfg)
        st = os.path.join(path, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu")
        sd = {k: v for k, v in sd.items() if not k.endswith("position_ids")}
        model.load_state_dict(sd, strict=True)
        return model.eval()

    @property
    def dtype(self):
        return self.visual_projection.weight.dtype

    @property
    def device(self):
        return self.visual_projection.weight.device

    def _pack(self, dt):
        return dict(proj=self.visual_projection.weight.detach().to(dt).contiguous())

    @torch.no_grad()
    def forward(self, pixel_values, **_):
        dt = self.compute_dtype or (self.dtype if self.dtype in (torch.float16, torch.bfloat16) else torch.float16)
        h, pooled = self.vision_model.run(dt, pixel_values)
        embeds = ops.gemm(pooled, self.packed(dt)["proj"], out_f32=True)
        out_dt = self.dtype if self.dtype in (torch.float16, torch.bfloat16) else torch.float32
        return SimpleNamespace(image_embeds=embeds.to(out_dt), last_hidden_state=h.to(out_dt))
