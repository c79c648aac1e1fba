"""ORACLE — TEST INFRASTRUCTURE ONLY (Tier-2: self-contained CPU fp32 restatement).

Restates, in plain PyTorch, the model graphs of the reference's denoising hot path so that the
oracle also runs where /root/reference is not mounted (the GPU box).  Module attribute paths
reproduce the reference's state-dict keys exactly, so the reference checkpoints
(`denoising_unet.pth`, `reference_unet.pth`, `pose_guider.pth`, `motion_module.pth`, SD1.5 unet)
load unchanged.  tests/test_oracle_vs_reference.py proves it equal to the reference's own
`src/models/*.py` (run behind oracle/diffusers_standin.py) on seeded weights; golden tensors
frozen from that run are under tests/golden/.

Reference files restated (paths relative to /root/reference):
  src/models/resnet.py                    InflatedConv3d/InflatedGroupNorm/ResnetBlock3D/Up-/Downsample3D
  src/models/attention.py:298-445         TemporalBasicTransformerBlock  (and :12-295 BasicTransformerBlock)
  src/models/mutual_self_attention.py     read/write bank semantics of the patched block forward
  src/models/transformer_3d.py:103-169    Transformer3DModel  (transformer_2d.py:213-396 per image)
  src/models/motion_module.py             VanillaTemporalModule ... VersatileAttention, PositionalEncoding
  src/models/unet_3d_blocks.py            down / mid / up blocks
  src/models/unet_3d_edit_bkfill.py       UNet3DConditionModel (+ from_pretrained_2d merge rules)
  src/models/unet_2d_condition.py         UNet2DConditionModel (conv_norm_out/conv_out removed :645-653,1295-1299)
  src/models/pose_guider.py               PoseGuider
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from einops import rearrange

from .primitives import Attention, FeedForward, TimestepEmbedding, Timesteps


# ---- src/models/resnet.py -------------------------------------------------------------------------
class InflatedConv3d(nn.Conv2d):  # resnet.py:9-17
    def forward(self, x):
        f = x.shape[2]
        x = rearrange(x, "b c f h w -> (b f) c h w")
        x = super().forward(x)
        return rearrange(x, "(b f) c h w -> b c f h w", f=f)


class InflatedGroupNorm(nn.GroupNorm):  # resnet.py:20-28 (statistics per frame)
    def forward(self, x):
        f = x.shape[2]
        x = rearrange(x, "b c f h w -> (b f) c h w")
        x = super().forward(x)
        return rearrange(x, "(b f) c h w -> b c f h w", f=f)


class Upsample3D(nn.Module):  # resnet.py:31-90
    def __init__(self, channels, out_channels=None):
        super().__init__()
        self.conv = InflatedConv3d(channels, out_channels or channels, 3, padding=1)

    def forward(self, x, output_size=None):
        if output_size is None:
            x = F.interpolate(x, scale_factor=[1.0, 2.0, 2.0], mode="nearest")
        else:
            x = F.interpolate(x, size=output_size, mode="nearest")
        return self.conv(x)


class Downsample3D(nn.Module):  # resnet.py:93-120
    def __init__(self, channels, out_channels=None, padding=1):
        super().__init__()
        self.conv = InflatedConv3d(channels, out_channels or channels, 3, stride=2, padding=padding)

    def forward(self, x):
        return self.conv(x)


class ResnetBlock3D(nn.Module):  # resnet.py:123-247
    def __init__(self, in_channels, out_channels, temb_channels, groups=32, eps=1e-5, output_scale_factor=1.0):
        super().__init__()
        self.norm1 = InflatedGroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = InflatedConv3d(in_channels, out_channels, 3, stride=1, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = InflatedGroupNorm(groups, out_channels, eps=eps, affine=True)
        self.conv2 = InflatedConv3d(out_channels, out_channels, 3, stride=1, padding=1)
        self.output_scale_factor = output_scale_factor
        self.conv_shortcut = InflatedConv3d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return (x + h) / self.output_scale_factor


# ---- src/models/attention.py + mutual_self_attention.py ---------------------------------------------
class SpatialTransformerBlock(nn.Module):
    """TemporalBasicTransformerBlock (attention.py:298-445) / BasicTransformerBlock (:12-295) with the
    patched forward of ReferenceAttentionControl (mutual_self_attention.py:93-276).  mode: None | 'write' | 'read'."""

    def __init__(self, dim, heads, head_dim, cross_attention_dim):
        super().__init__()
        self.attn1 = Attention(query_dim=dim, heads=heads, dim_head=head_dim)
        self.norm1 = nn.LayerNorm(dim)
        self.attn2 = Attention(query_dim=dim, cross_attention_dim=cross_attention_dim, heads=heads, dim_head=head_dim)
        self.norm2 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim, activation_fn="geglu")
        self.norm3 = nn.LayerNorm(dim)
        self.mode = None
        self.bank = []
        self.do_cfg = True

    def forward(self, x, encoder_hidden_states, video_length=1):
        n = self.norm1(x)
        if self.mode == "write":  # :137-147
            self.bank.append(n.clone())
            x = self.attn1(n) + x
        elif self.mode == "read":  # :148-197
            bank_fea = [rearrange(d.unsqueeze(1).repeat(1, video_length, 1, 1), "b t l c -> (b t) l c") for d in self.bank]
            kv = torch.cat([n] + bank_fea, dim=1)
            x_uc = self.attn1(n, encoder_hidden_states=kv) + x
            if self.do_cfg:
                half = x.shape[0] // 2  # rows [0, half) are the unconditional batch element(s)
                x_c = x_uc.clone()
                x_c[:half] = self.attn1(n[:half], encoder_hidden_states=n[:half]) + x[:half]
                x = x_c
            else:
                x = x_uc
        else:
            x = self.attn1(n) + x
        x = self.attn2(self.norm2(x), encoder_hidden_states=encoder_hidden_states) + x
        return self.ff(self.norm3(x)) + x


class Transformer3DModel(nn.Module):  # transformer_3d.py:27-169 (use_linear_projection=False)
    def __init__(self, heads, head_dim, in_channels, cross_attention_dim, groups=32):
        super().__init__()
        inner = heads * head_dim
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([SpatialTransformerBlock(inner, heads, head_dim, cross_attention_dim)])
        self.proj_out = nn.Conv2d(inner, in_channels, 1)

    def forward(self, x, encoder_hidden_states):
        f = x.shape[2]
        x = rearrange(x, "b c f h w -> (b f) c h w")
        if encoder_hidden_states.shape[0] != x.shape[0]:
            encoder_hidden_states = encoder_hidden_states.repeat_interleave(f, dim=0)  # "b n c -> (b f) n c"
        b, c, h, w = x.shape
        res = x
        y = self.proj_in(self.norm(x))
        y = y.permute(0, 2, 3, 1).reshape(b, h * w, -1)
        for blk in self.transformer_blocks:
            y = blk(y, encoder_hidden_states, video_length=f)
        y = y.reshape(b, h, w, -1).permute(0, 3, 1, 2).contiguous()
        y = self.proj_out(y) + res
        return rearrange(y, "(b f) c h w -> b c f h w", f=f)


# ---- src/models/motion_module.py ----------------------------------------------------------------------
class PositionalEncoding(nn.Module):  # motion_module.py:264-279
    def __init__(self, d_model, max_len=24):
        super().__init__()
        position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(1, max_len, d_model)
        pe[0, :, 0::2] = torch.sin(position * div_term)
        pe[0, :, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe)

    def forward(self, x):
        return x + self.pe[:, : x.size(1)]


class VersatileAttention(Attention):  # motion_module.py:282-390 (Temporal_Self)
    def __init__(self, dim, heads, head_dim, max_len):
        super().__init__(query_dim=dim, heads=heads, dim_head=head_dim)
        self.pos_encoder = PositionalEncoding(dim, max_len=max_len)

    def forward(self, x, video_length):
        d = x.shape[1]
        x = rearrange(x, "(b f) d c -> (b d) f c", f=video_length)
        x = self.pos_encoder(x)
        x = self.processor(self, x)
        return rearrange(x, "(b d) f c -> (b f) d c", d=d)


class TemporalTransformerBlock(nn.Module):  # motion_module.py:187-261
    def __init__(self, dim, heads, head_dim, max_len):
        super().__init__()
        self.attention_blocks = nn.ModuleList([VersatileAttention(dim, heads, head_dim, max_len) for _ in range(2)])
        self.norms = nn.ModuleList([nn.LayerNorm(dim) for _ in range(2)])
        self.ff = FeedForward(dim, activation_fn="geglu")
        self.ff_norm = nn.LayerNorm(dim)

    def forward(self, x, video_length):
        for attn, norm in zip(self.attention_blocks, self.norms):
            x = attn(norm(x), video_length) + x
        return self.ff(self.ff_norm(x)) + x


class TemporalTransformer3DModel(nn.Module):  # motion_module.py:94-184
    def __init__(self, in_channels, heads, head_dim, max_len, groups=32):
        super().__init__()
        inner = heads * head_dim
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([TemporalTransformerBlock(inner, heads, head_dim, max_len)])
        self.proj_out = nn.Linear(inner, in_channels)

    def forward(self, x):
        f = x.shape[2]
        x = rearrange(x, "b c f h w -> (b f) c h w")
        b, c, h, w = x.shape
        res = x
        y = self.norm(x).permute(0, 2, 3, 1).reshape(b, h * w, c)
        y = self.proj_in(y)
        for blk in self.transformer_blocks:
            y = blk(y, video_length=f)
        y = self.proj_out(y).reshape(b, h, w, c).permute(0, 3, 1, 2).contiguous()
        return rearrange(y + res, "(b f) c h w -> b c f h w", f=f)


class VanillaTemporalModule(nn.Module):  # motion_module.py:44-91
    def __init__(self, in_channels, heads=8, max_len=32):
        super().__init__()
        # norm_num_groups is NOT forwarded by get_motion_module: always the default 32 (motion_module.py:108)
        self.temporal_transformer = TemporalTransformer3DModel(in_channels, heads, in_channels // heads, max_len, 32)
        nn.init.zeros_(self.temporal_transformer.proj_out.weight)  # zero_module (:72-75)
        nn.init.zeros_(self.temporal_transformer.proj_out.bias)

    def forward(self, x):
        return self.temporal_transformer(x)


# ---- src/models/unet_3d_blocks.py --------------------------------------------------------------------
class DownBlock(nn.Module):  # CrossAttnDownBlock3D :296-464 / DownBlock3D :467-583
    def __init__(self, cin, cout, temb, layers, attn, heads, cross_dim, add_down, motion, groups, eps, mm_kw):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock3D(cin if i == 0 else cout, cout, temb, groups, eps) for i in range(layers)])
        if attn:
            self.attentions = nn.ModuleList([Transformer3DModel(heads, cout // heads, cout, cross_dim, groups) for _ in range(layers)])
        self.has_attn = attn
        self.motion_modules = nn.ModuleList([VanillaTemporalModule(cout, **mm_kw) if motion else None for _ in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample3D(cout, cout)]) if add_down else None

    def forward(self, x, temb, ehs):
        outs = ()
        for i, res in enumerate(self.resnets):
            x = res(x, temb)
            if self.has_attn:
                x = self.attentions[i](x, ehs)
            if self.motion_modules[i] is not None:
                x = self.motion_modules[i](x)
            outs += (x,)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs += (x,)
        return x, outs


class MidBlock(nn.Module):  # UNetMidBlock3DCrossAttn :170-293
    def __init__(self, ch, temb, heads, cross_dim, motion, groups, eps, mm_kw):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock3D(ch, ch, temb, groups, eps), ResnetBlock3D(ch, ch, temb, groups, eps)])
        self.attentions = nn.ModuleList([Transformer3DModel(heads, ch // heads, ch, cross_dim, groups)])
        self.motion_modules = nn.ModuleList([VanillaTemporalModule(ch, **mm_kw) if motion else None])

    def forward(self, x, temb, ehs):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, ehs)
        if self.motion_modules[0] is not None:
            x = self.motion_modules[0](x)
        return self.resnets[1](x, temb)


class UpBlock(nn.Module):  # CrossAttnUpBlock3D :586-745 / UpBlock3D :748-862
    def __init__(self, cin, cout, prev, temb, layers, attn, heads, cross_dim, add_up, motion, groups, eps, mm_kw):
        super().__init__()
        self.resnets = nn.ModuleList()
        for i in range(layers):
            skip = cin if i == layers - 1 else cout
            rin = prev if i == 0 else cout
            self.resnets.append(ResnetBlock3D(rin + skip, cout, temb, groups, eps))
        if attn:
            self.attentions = nn.ModuleList([Transformer3DModel(heads, cout // heads, cout, cross_dim, groups) for _ in range(layers)])
        self.has_attn = attn
        self.motion_modules = nn.ModuleList([VanillaTemporalModule(cout, **mm_kw) if motion else None for _ in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample3D(cout, cout)]) if add_up else None

    def forward(self, x, skips, temb, ehs, upsample_size=None):
        for i, res in enumerate(self.resnets):
            x = torch.cat([x, skips[-1]], dim=1)
            skips = skips[:-1]
            x = res(x, temb)
            if self.has_attn:
                x = self.attentions[i](x, ehs)
            if self.motion_modules[i] is not None:
                x = self.motion_modules[i](x)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x, upsample_size)
        return x


# ---- src/models/unet_3d_edit_bkfill.py / unet_2d_condition.py ---------------------------------------
SD15_UNET_CONFIG = dict(block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, attention_head_dim=8,
                        cross_attention_dim=768, norm_num_groups=32, norm_eps=1e-5)


class UNetBase(nn.Module):
    """SD1.5 topology; `motion` adds the AnimateDiff motion modules (denoising UNet), `with_out` the output head."""

    def __init__(self, in_channels, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                 attention_head_dim=8, cross_attention_dim=768, norm_num_groups=32, norm_eps=1e-5, motion=False,
                 with_out=True, temporal_position_encoding_max_len=32, motion_heads=8):
        super().__init__()
        boc = list(block_out_channels)
        temb = boc[0] * 4
        heads = attention_head_dim  # SD1.5: attention_head_dim is the NUMBER of heads
        g, eps = norm_num_groups, norm_eps
        mm_kw = dict(heads=motion_heads, max_len=temporal_position_encoding_max_len)
        self.conv_in = InflatedConv3d(in_channels, boc[0], 3, padding=1)
        self.time_proj = Timesteps(boc[0], True, 0)
        self.time_embedding = TimestepEmbedding(boc[0], temb)
        self.down_blocks = nn.ModuleList()
        out = boc[0]
        for i, co in enumerate(boc):
            cin, out = out, co
            last = i == len(boc) - 1
            self.down_blocks.append(DownBlock(cin, out, temb, layers_per_block, not last, heads, cross_attention_dim,
                                              not last, motion, g, eps, mm_kw))
        self.mid_block = MidBlock(boc[-1], temb, heads, cross_attention_dim, motion, g, eps, mm_kw)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        out = rev[0]
        for i, co in enumerate(rev):
            prev, out = out, co
            cin = rev[min(i + 1, len(boc) - 1)]
            last = i == len(boc) - 1
            self.up_blocks.append(UpBlock(cin, out, prev, temb, layers_per_block + 1, i != 0, heads, cross_attention_dim,
                                          not last, motion, g, eps, mm_kw))
        self.num_upsamplers = len(boc) - 1
        if with_out:
            self.conv_norm_out = InflatedGroupNorm(g, boc[0], eps=eps)
            self.conv_out = InflatedConv3d(boc[0], out_channels, 3, padding=1)
        self.with_out = with_out

    def spatial_blocks(self):
        """The 16 spatial transformer blocks sorted like mutual_self_attention.py:295-297 (stable, by -dim)."""
        # registration order in the reference is down_blocks, up_blocks, mid_block (mid_block is first set to
        # None, i.e. a plain attribute, and only becomes a registered child after up_blocks exists:
        # unet_3d_edit_bkfill.py:109-111,158; unet_2d_condition.py:455-456,531)
        blocks = [m for part in (self.down_blocks, self.up_blocks, self.mid_block) for m in part.modules()
                  if isinstance(m, SpatialTransformerBlock)]
        return sorted(blocks, key=lambda b: -b.norm1.normalized_shape[0])

    def forward(self, sample, timestep, encoder_hidden_states, pose_cond_fea=None):
        up_factor = 2 ** self.num_upsamplers
        forward_upsample_size = any(s % up_factor != 0 for s in sample.shape[-2:])
        t = torch.as_tensor(timestep, device=sample.device)
        if t.dim() == 0:
            t = t[None]
        t = t.expand(sample.shape[0])
        emb = self.time_embedding(self.time_proj(t).to(sample.dtype))
        x = self.conv_in(sample)
        if pose_cond_fea is not None:
            x = x + pose_cond_fea
        skips = (x,)
        for blk in self.down_blocks:
            x, outs = blk(x, emb, encoder_hidden_states)
            skips += outs
        x = self.mid_block(x, emb, encoder_hidden_states)
        for i, blk in enumerate(self.up_blocks):
            n = len(blk.resnets)
            res, skips = skips[-n:], skips[:-n]
            size = skips[-1].shape[2:] if (i != len(self.up_blocks) - 1 and forward_upsample_size) else None
            x = blk(x, res, emb, encoder_hidden_states, size)
        if self.with_out:
            x = self.conv_out(F.silu(self.conv_norm_out(x)))
        return x


class UNet3DConditionModel(UNetBase):
    """Denoising UNet: 8 input channels (4 noisy + 4 background latents), pose add, motion modules."""

    def __init__(self, **kw):
        kw.setdefault("in_channels", 8)
        super().__init__(motion=True, with_out=True, **kw)


class UNet2DConditionModel(UNetBase):
    """Reference UNet: plain SD1.5 without conv_norm_out/conv_out; run on [b,4,1,h,w] (per-image math)."""

    def __init__(self, **kw):
        kw.setdefault("in_channels", 4)
        super().__init__(motion=False, with_out=False, **kw)

    def forward(self, sample, timestep, encoder_hidden_states):
        if sample.dim() == 4:
            sample = sample[:, :, None]
        return super().forward(sample, timestep, encoder_hidden_states)[:, :, 0]


def denoising_state_dict_from_sd15(sd15, motion_sd):
    """from_pretrained_2d merge (unet_3d_edit_bkfill.py:639-674): SD1.5 weights + motion module, conv_in zero-padded 4 -> 8."""
    sd = dict(sd15)
    sd.update(motion_sd)
    w = sd["conv_in.weight"]
    if w.shape[1] != 8:
        sd["conv_in.weight"] = torch.cat([w, torch.zeros(w.shape[0], 8 - w.shape[1], *w.shape[2:], dtype=w.dtype)], dim=1)
    return sd


class ReferenceAttentionControl:
    """mutual_self_attention.py:19-374, fusion_blocks='full'."""

    def __init__(self, unet, mode, do_classifier_free_guidance=True, **_):
        self.unet, self.mode = unet, mode
        for b in unet.spatial_blocks():
            b.mode, b.bank, b.do_cfg = mode, [], do_classifier_free_guidance

    def update(self, writer, dtype=torch.float16):
        # banks are cast to fp16 whatever the model dtype (:313,349); torch.cat promotes back
        for r, w in zip(self.unet.spatial_blocks(), writer.unet.spatial_blocks()):
            r.bank = [v.clone().to(dtype) for v in w.bank]

    def clear(self):
        for b in self.unet.spatial_blocks():
            b.bank.clear()


# ---- src/models/pose_guider.py -------------------------------------------------------------------------
class PoseGuider(nn.Module):
    def __init__(self, conditioning_embedding_channels=320, conditioning_channels=3, block_out_channels=(16, 32, 96, 256)):
        super().__init__()
        boc = block_out_channels
        self.conv_in = InflatedConv3d(conditioning_channels, boc[0], 3, padding=1)
        self.blocks = nn.ModuleList()
        for i in range(len(boc) - 1):
            self.blocks.append(InflatedConv3d(boc[i], boc[i], 3, padding=1))
            self.blocks.append(InflatedConv3d(boc[i], boc[i + 1], 3, padding=1, stride=2))
        self.conv_out = InflatedConv3d(boc[-1], conditioning_embedding_channels, 3, padding=1)
        nn.init.zeros_(self.conv_out.weight)
        nn.init.zeros_(self.conv_out.bias)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    def forward(self, c):
        x = F.silu(self.conv_in(c))
        for b in self.blocks:
            x = F.silu(b(x))
        return self.conv_out(x)
