"""Denoising UNet (3-D, motion modules, pose add, reference-bank attention) and reference UNet
(2-D, bank writer) on the HIP kernels.  Public surface mirrors the reference:

  UNet3DConditionModel   src/models/unet_3d_edit_bkfill.py:30-682  (ctor kwargs, forward(), from_pretrained_2d())
  UNet2DConditionModel   src/models/unet_2d_condition.py           (ctor kwargs, forward(), from_pretrained())
  ReferenceAttentionControl  src/models/mutual_self_attention.py:19-374 (write / read / update / clear)

State-dict keys are those of the reference classes (1274 / 682 entries at SD1.5 size).
"""
import json
from pathlib import Path

import torch
import torch.nn as nn

from . import ops
from .modules import (Ctx, Downsample, EarlyExit, HipModule, MotionModule, ResnetBlock, SpatialTransformer,
                      SpatialTransformerBlock, Upsample, _f32)
from .packing import pack_conv, pack_conv_taps, pad_vec


class _Linear2(nn.Module):  # diffusers TimestepEmbedding key layout
    def __init__(self, cin, cout):
        super().__init__()
        self.linear_1 = nn.Linear(cin, cout)
        self.linear_2 = nn.Linear(cout, cout)


class DownBlock(nn.Module):  # CrossAttnDownBlock3D / DownBlock3D, src/models/unet_3d_blocks.py:296-583
    def __init__(self, cin, cout, temb, layers, attn, heads, cross_dim, add_down, motion, groups, eps, mm_kw):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock(cin if i == 0 else cout, cout, temb, groups, eps) for i in range(layers)])
        if attn:
            self.attentions = nn.ModuleList([SpatialTransformer(heads, cout // heads, cout, cross_dim, groups) for _ in range(layers)])
        self.has_attn = attn
        if motion:
            self.motion_modules = nn.ModuleList([MotionModule(cout, **mm_kw) for _ in range(layers)])
        self.has_motion = motion
        self.downsamplers = nn.ModuleList([Downsample(cout, cout)]) if add_down else None

    def run(self, ctx, x):
        outs = []
        for i, res in enumerate(self.resnets):
            x = res.run(ctx, x)
            if self.has_attn:
                x = self.attentions[i].run(ctx, x)
            if self.has_motion:
                x = self.motion_modules[i].run(ctx, x)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0].run(ctx, x)
            outs.append(x)
        return x, outs


class MidBlock(nn.Module):  # UNetMidBlock3DCrossAttn, src/models/unet_3d_blocks.py:170-293
    def __init__(self, ch, temb, heads, cross_dim, motion, groups, eps, mm_kw):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock(ch, ch, temb, groups, eps), ResnetBlock(ch, ch, temb, groups, eps)])
        self.attentions = nn.ModuleList([SpatialTransformer(heads, ch // heads, ch, cross_dim, groups)])
        if motion:
            self.motion_modules = nn.ModuleList([MotionModule(ch, **mm_kw)])
        self.has_motion = motion

    def run(self, ctx, x):
        x = self.resnets[0].run(ctx, x)
        x = self.attentions[0].run(ctx, x)
        if self.has_motion:
            x = self.motion_modules[0].run(ctx, x)
        return self.resnets[1].run(ctx, x)


class UpBlock(nn.Module):  # CrossAttnUpBlock3D / UpBlock3D, src/models/unet_3d_blocks.py:586-862
    def __init__(self, cin, cout, prev, temb, layers, attn, heads, cross_dim, add_up, motion, groups, eps, mm_kw):
        super().__init__()
        self.resnets = nn.ModuleList()
        for i in range(layers):
            skip = cin if i == layers - 1 else cout
            rin = prev if i == 0 else cout
            self.resnets.append(ResnetBlock(rin + skip, cout, temb, groups, eps))
        if attn:
            self.attentions = nn.ModuleList([SpatialTransformer(heads, cout // heads, cout, cross_dim, groups) for _ in range(layers)])
        self.has_attn = attn
        if motion:
            self.motion_modules = nn.ModuleList([MotionModule(cout, **mm_kw) for _ in range(layers)])
        self.has_motion = motion
        self.upsamplers = nn.ModuleList([Upsample(cout, cout)]) if add_up else None

    def run(self, ctx, x, skips, upsample_size=None):
        for i, res in enumerate(self.resnets):
            x = res.run(ctx, x, skip=skips.pop())  # torch.cat([h, res], dim=1) is never materialised
            if self.has_attn:
                x = self.attentions[i].run(ctx, x)
            if self.has_motion:
                x = self.motion_modules[i].run(ctx, x)
        if self.upsamplers is not None:
            x = self.upsamplers[0].run(ctx, x, upsample_size)
        return x


class _Config(dict):
    __getattr__ = dict.__getitem__


class UNetBase(HipModule):
    def __init__(self, in_channels, out_channels, block_out_channels, layers_per_block, attention_head_dim,
                 cross_attention_dim, norm_num_groups, norm_eps, motion, with_out, max_len, motion_heads):
        super().__init__()
        boc = list(block_out_channels)
        temb = boc[0] * 4
        heads = attention_head_dim  # SD1.5: "attention_head_dim" = number of heads
        g, eps = norm_num_groups, norm_eps
        mm_kw = dict(heads=motion_heads, max_len=max_len)
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.time_embedding = _Linear2(boc[0], temb)
        self.down_blocks = nn.ModuleList()
        out = boc[0]
        for i, co in enumerate(boc):
            cin, out = out, co
            last = i == len(boc) - 1
            self.down_blocks.append(DownBlock(cin, out, temb, layers_per_block, not last, heads, cross_attention_dim,
                                              not last, motion, g, eps, mm_kw))
        # registered after up_blocks in the reference; order here only matters for bank pairing, done by name below
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
        if with_out:
            self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=eps)
            self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)
        self.with_out, self.groups, self.eps = with_out, g, eps
        self.in_channels, self.out_channels, self.boc = in_channels, out_channels, boc
        self.num_upsamplers = len(boc) - 1
        self.cross_dim = cross_attention_dim
        self.compute_dtype = torch.float16
        # "half": 16-bit MFMA operands, the fused kernels (the benchmarked path).  "split": both operands of every conv / Linear
        # as hi + lo pairs through the same GEMM kernels, unfused (mimo_amd.precise): ~3x the MFMA work, meets the 1e-3 bar where
        # the 16-bit policy sits at 1.1-1.4e-3 (configs[0], guidance-3.5 cases of the test models) and with bf16
        self.precision = "half"
        # column slices into the per-forward fused time-embedding / cross-attention matrices
        off = 0
        for m in self.modules():
            if isinstance(m, ResnetBlock):
                m.temb_slice = (off, off + m.out_channels)
                off += m.out_channels
        off = 0
        for m in self.spatial_blocks():
            m.attn2_slice = (off, off + m.dim)
            off += m.dim
        if with_out:   # (see ResnetBlock.run and ops.EDGE_SPLIT bit 3: the last two resnets in front of the output head)
            for r in self.up_blocks[-1].resnets[-2:]:
                r.edge_parts = ("sc", "conv2")

    # ---- plumbing ----
    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        if self.conv_in.weight.dtype in (torch.float16, torch.bfloat16):
            self.compute_dtype = self.conv_in.weight.dtype  # .to(dtype=fp16) selects the MFMA operand type
        return r

    def spatial_blocks(self):
        """Transformer blocks in the reference's pairing order (mutual_self_attention.py:295-297,342-347):
        DFS over down, up, mid, stably sorted by descending channel count."""
        blocks = [m for part in (self.down_blocks, self.up_blocks, self.mid_block) for m in part.modules()
                  if isinstance(m, SpatialTransformerBlock)]
        return sorted(blocks, key=lambda b: -b.dim)

    def _pack(self, dt):
        dev = self.device
        h = lambda w: w.detach().to(dt).contiguous()
        cin_pad = (self.in_channels + 7) // 8 * 8
        d = dict(ci_w=pack_conv(self.conv_in.weight, dt, cin_pad=cin_pad), ci_b=_f32(self.conv_in.bias),
                 t1_w=h(self.time_embedding.linear_1.weight), t1_b=_f32(self.time_embedding.linear_1.bias),
                 t2_w=h(self.time_embedding.linear_2.weight), t2_b=_f32(self.time_embedding.linear_2.bias))
        res = [m for m in self.modules() if isinstance(m, ResnetBlock)]
        d["temb_w"] = torch.cat([m.time_emb_proj.weight for m in res], 0).detach().to(dt).contiguous()
        d["temb_b"] = torch.cat([m.time_emb_proj.bias for m in res], 0).detach().float().contiguous()
        mats = [m.attn2_matrix() for m in self.spatial_blocks()]
        d["a2_w"] = torch.cat([w for w, _ in mats], 0).to(dt).contiguous()
        d["a2_b"] = torch.cat([b for _, b in mats], 0).contiguous()
        if self.with_out:
            cout_pad = (self.out_channels + 3) // 4 * 4
            d.update(no_g=_f32(self.conv_norm_out.weight), no_b=_f32(self.conv_norm_out.bias),
                     co_w=pack_conv(self.conv_out.weight, dt, cout_pad=cout_pad), co_b=pad_vec(self.conv_out.bias, cout_pad),
                     co_wt=pack_conv_taps(self.conv_out.weight, dt, cout_pad=cout_pad))
        half = self.boc[0] // 2
        import math
        d["freqs"] = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=dev) / half)
        return d

    def timestep_table(self, timesteps, b):
        """Sinusoidal embeddings (Timesteps(flip_sin_to_cos, shift 0)) of a clip's timesteps as ONE half tensor
        [len(timesteps), b, C0], built on the host and uploaded once: run_tokens(t_emb=table[i]) then launches no
        elementwise glue per step.  Same formula and fp32 arithmetic as the per-call path of _time_and_cross — but THIS table
        takes exp / cos / sin from the host's libm (as the CPU oracle does) and the per-call path from the device's: the two
        can differ by one fp32 ulp before the rounding to half, i.e. by half an ulp in a few of the 320 entries (8 at
        t = 999).  The pipeline — eager or replaying a hipGraph — always goes through the tables (clip_tables), so its runs
        agree bit for bit; only a direct run_tokens(x, timestep, ...) call takes the device formula."""
        import math
        half = self.boc[0] // 2
        freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
        ang = torch.tensor([float(t) for t in timesteps], dtype=torch.float32)[:, None] * freqs[None, :]
        emb = torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1).to(self.compute_dtype)
        return emb[:, None, :].repeat(1, b, 1).contiguous().to(self.device)

    def clip_tables(self, timesteps, ehs, b):
        """Everything of a forward that depends only on (timestep, encoder_hidden_states), for ALL steps of a clip at once:
        the time-embedding chain (Timesteps -> TimestepEmbedding -> SiLU -> the 22 stacked time_emb_proj, three GEMMs over
        len(timesteps) * b rows instead of three M = 2 launches per forward) and the 16 collapsed cross-attentions (one
        GEMM per clip: they do not depend on the step).  Returns (temb fp32 [S, b, sum(Cout)], attn2 fp32 [b, sum(C)]);
        run_tokens(temb=..., attn2=...) then starts at conv_in.  The arithmetic per row is that of _time_and_cross."""
        if self.precision == "split" or (ops.EDGE_SPLIT & 2):
            from . import precise
            return precise.clip_tables(self, timesteps, ehs, b)
        dt = self.compute_dtype
        p = self.packed(dt)
        S = len(timesteps)
        t_emb = self.timestep_table(timesteps, b).view(S * b, -1)
        e1 = ops.gemm(t_emb, p["t1_w"], bias=p["t1_b"], silu=True)
        emb = ops.gemm(e1, p["t2_w"], bias=p["t2_b"], silu=True)
        temb = ops.gemm(emb, p["temb_w"], bias=p["temb_b"], out_f32=True).view(S, b, -1)
        e = ehs.reshape(b, -1).to(device=self.device, dtype=dt).contiguous()
        return temb, ops.gemm(e, p["a2_w"], bias=p["a2_b"], out_f32=True)

    def _time_and_cross(self, ctx, p, timestep, ehs):
        """Timesteps(flip_sin_to_cos, shift 0) -> TimestepEmbedding -> silu -> ALL 22 time_emb_proj in one GEMM;
        ALL 16 collapsed cross-attentions in one GEMM (src/models/unet_3d_edit_bkfill.py:447-468, resnet.py:226)."""
        dev = self.device
        if ctx.temb is not None and ctx.attn2 is not None:  # rows of clip_tables(): nothing left to do per forward
            return
        if ctx.t_emb is not None:  # precomputed row of timestep_table()
            t_emb = ctx.t_emb
            assert t_emb.shape == (ctx.b, self.boc[0]) and t_emb.dtype == ctx.dtype and t_emb.is_contiguous()
        else:
            if torch.is_tensor(timestep) and timestep.device == dev:
                t = timestep.reshape(-1).float().expand(ctx.b)  # device scalar: capturable in a hipGraph
            else:
                t = torch.as_tensor(timestep, device=dev).reshape(-1).float().expand(ctx.b)
            ang = t[:, None] * p["freqs"][None, :]
            t_emb = torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1).to(ctx.dtype)  # [b, 320] (glue, b x 320)
        e1 = ops.gemm(t_emb, p["t1_w"], bias=p["t1_b"], silu=True)
        emb = ops.gemm(e1, p["t2_w"], bias=p["t2_b"], silu=True)  # silu(emb): every consumer applies the nonlinearity first
        ctx.temb = ops.gemm(emb, p["temb_w"], bias=p["temb_b"], out_f32=True)
        e = ehs.reshape(ctx.b, -1).to(device=dev, dtype=ctx.dtype).contiguous()
        ctx.attn2 = ops.gemm(e, p["a2_w"], bias=p["a2_b"], out_f32=True)

    def run_tokens(self, x_tok, timestep, ehs, b, F, pose_tok=None, ctx=None, t_emb=None, temb=None, attn2=None):
        """x_tok: half [b*F, h, w, Cin_pad8]; ehs: [b, 1, 768]; pose_tok: [b*F, h, w, C0] (fp32|half) or None.
        Returns fp32 tokens [b*F, h, w, Cout_pad4] (or the last hidden state when there is no output head).
        precision "split": x_tok may be fp32 (then nothing of the input is rounded); see mimo_amd.precise."""
        if self.precision == "split":
            from . import precise
            return precise.run_tokens(self, x_tok, timestep, ehs, b, F, pose_tok, ctx, temb, attn2)
        dt = self.compute_dtype
        p = self.packed(dt)
        ctx = ctx or Ctx(dt, b, F)
        ctx.t_emb = t_emb
        if temb is not None and attn2 is not None:
            assert temb.shape[0] == b and attn2.shape[0] == b and temb.stride(1) == 1 and attn2.stride(1) == 1
            ctx.temb, ctx.attn2 = temb, attn2
        self._time_and_cross(ctx, p, timestep, ehs)
        n, H, W, _ = x_tok.shape
        up = 2 ** self.num_upsamplers
        forward_upsample_size = (H % up != 0) or (W % up != 0)
        if (ops.EDGE_SPLIT & 4) or x_tok.dtype == torch.float32:   # the input convolution with split operands (fp32 tokens: nothing rounded)
            from . import precise
            P = self.packed_split(dt, lambda d: precise._unet_pack(self, d))
            x32 = x_tok if x_tok.dtype == torch.float32 else x_tok.float()
            x = ops.conv2d(ops.split3(x32.contiguous(), dtype=dt, ld=P["ci"].shape[1] // 9), P["ci"], self.boc[0], bias=p["ci_b"],
                           residual=None if pose_tok is None else pose_tok.float(), out_f32=True)
        else:
            x = ops.conv2d(x_tok, p["ci_w"], self.boc[0], bias=p["ci_b"], residual=pose_tok, out_f32=True, colstats=True)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk.run(ctx, x)
            skips += outs
        x = self.mid_block.run(ctx, x)
        for i, blk in enumerate(self.up_blocks):
            nres = len(blk.resnets)
            res, skips = skips[-nres:], skips[:-nres]
            size = None
            if i != len(self.up_blocks) - 1 and forward_upsample_size:
                size = skips[-1].shape[1:3]
            x = blk.run(ctx, x, res, size)
        if not self.with_out:
            return x
        if ops.EDGE_SPLIT & 1:   # the output head with split operands (its rounding lands on the prediction directly)
            from . import precise
            return precise.output_head(self, x, self.packed_split(dt, lambda d: precise._unet_pack(self, d)), p)
        a, _ = ops.group_norm(x, p["no_g"], p["no_b"], groups=self.groups, eps=self.eps, silu=True, dtype=dt)
        if ops.THIN_OUT and p["co_w"].shape[0] <= 16:
            return ops.conv3x3_thin_out(a, p["co_wt"], p["co_w"].shape[0], bias=p["co_b"])
        return ops.conv2d(a, p["co_w"], p["co_w"].shape[0], bias=p["co_b"], out_f32=True)


MM_DEFAULT = dict(num_attention_heads=8, temporal_position_encoding_max_len=32)


class UNet3DConditionModel(UNetBase):
    """Drop-in for src/models/unet_3d_edit_bkfill.py::UNet3DConditionModel (SD1.5 config + inference_v2.yaml kwargs)."""

    def __init__(self, sample_size=None, in_channels=8, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                 layers_per_block=2, norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=768, attention_head_dim=8,
                 use_inflated_groupnorm=True, use_motion_module=True, motion_module_resolutions=(1, 2, 4, 8),
                 motion_module_mid_block=True, motion_module_decoder_only=False, motion_module_type="Vanilla",
                 motion_module_kwargs=None, unet_use_cross_frame_attention=False, unet_use_temporal_attention=False,
                 **unused):
        mm = dict(MM_DEFAULT, **(motion_module_kwargs or {}))
        if not (use_inflated_groupnorm and use_motion_module and motion_module_mid_block and not motion_module_decoder_only
                and tuple(motion_module_resolutions) == (1, 2, 4, 8) and motion_module_type == "Vanilla"
                and not unet_use_cross_frame_attention and not unet_use_temporal_attention):
            raise NotImplementedError("only the configs/inference/inference_v2.yaml UNet variant is implemented")
        super().__init__(8, out_channels, block_out_channels, layers_per_block, attention_head_dim, cross_attention_dim,
                         norm_num_groups, norm_eps, True, True, mm["temporal_position_encoding_max_len"],
                         mm["num_attention_heads"])  # in_channels is forced to 8 (unet_3d_edit_bkfill.py:87)
        self.config = _Config(sample_size=sample_size, in_channels=8, out_channels=out_channels,
                              block_out_channels=list(block_out_channels), cross_attention_dim=cross_attention_dim)

    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, pose_cond_fea=None,
                attention_mask=None, down_block_additional_residuals=None, mid_block_additional_residual=None,
                return_dict=True):
        """sample [b,8,f,h,w], pose_cond_fea [b,320,f,h,w] -> (sample [b,4,f,h,w],) like the reference."""
        b, c, f, h, w = sample.shape
        dt = self.compute_dtype
        if self.precision == "split" or (ops.EDGE_SPLIT & 4):   # fp32 tokens: a layout copy, nothing is rounded
            x = torch.zeros((b * f, h, w, 8), device=sample.device, dtype=torch.float32)
            x[..., :c] = sample.float().permute(0, 2, 3, 4, 1).reshape(b * f, h, w, c)
            pose = None if pose_cond_fea is None else pose_cond_fea.float().permute(0, 2, 3, 4, 1).reshape(b * f, h, w, -1).contiguous()
        else:
            x = ops.ncfhw_to_tokens(sample.contiguous(), dt, cpad=8)
            pose = None if pose_cond_fea is None else ops.ncfhw_to_tokens(pose_cond_fea.contiguous(), dt)
        y = self.run_tokens(x, timestep, encoder_hidden_states, b, f, pose)
        out = ops.tokens_to_ncfhw(y, b, self.out_channels, f, h, w).to(sample.dtype)
        return _Out(sample=out) if return_dict else (out,)

    @classmethod
    def from_pretrained_2d(cls, pretrained_model_path, motion_module_path, subfolder=None, unet_additional_kwargs=None,
                           mm_zero_proj_out=False):
        """SD1.5 2-D weights + motion module, conv_in zero-padded 4 -> 8 (unet_3d_edit_bkfill.py:578-682)."""
        path = Path(pretrained_model_path)
        if subfolder is not None:
            path = path / subfolder
        cfg_file = path / "config.json"
        if not cfg_file.is_file():
            raise RuntimeError(f"{cfg_file} does not exist or is not a file")
        cfg = json.loads(cfg_file.read_text())
        names = ("sample_size", "out_channels", "block_out_channels", "layers_per_block", "norm_num_groups", "norm_eps",
                 "cross_attention_dim", "attention_head_dim")
        model = cls(**{k: cfg[k] for k in names if k in cfg}, **dict(unet_additional_kwargs or {}))
        sd = _load_weights(path)
        mm_path = Path(motion_module_path)
        if mm_path.is_file():
            if mm_path.suffix.lower() in (".pth", ".pt", ".ckpt"):
                msd = torch.load(mm_path, map_location="cpu", weights_only=True)
            elif mm_path.suffix.lower() == ".safetensors":
                from safetensors.torch import load_file
                msd = load_file(str(mm_path), device="cpu")
            else:
                raise RuntimeError(f"unknown file format for motion module weights: {mm_path.suffix}")
            if mm_zero_proj_out:
                msd = {k: v for k, v in msd.items() if "proj_out" not in k}
            sd.update(msd)
        w = sd["conv_in.weight"]
        if w.shape[1] != 8:
            sd["conv_in.weight"] = torch.cat([w, torch.zeros(w.shape[0], 8 - w.shape[1], *w.shape[2:], dtype=w.dtype)], dim=1)
        model.load_state_dict(sd, strict=False)
        return model


class UNet2DConditionModel(UNetBase):
    """Drop-in for src/models/unet_2d_condition.py::UNet2DConditionModel as the reference UNet (no conv_out)."""

    def __init__(self, sample_size=None, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                 layers_per_block=2, norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=768, attention_head_dim=8,
                 **unused):
        super().__init__(in_channels, out_channels, block_out_channels, layers_per_block, attention_head_dim,
                         cross_attention_dim, norm_num_groups, norm_eps, False, False, 32, 8)
        self.config = _Config(sample_size=sample_size, in_channels=in_channels, block_out_channels=list(block_out_channels),
                              cross_attention_dim=cross_attention_dim)

    def forward(self, sample, timestep, encoder_hidden_states, return_dict=True, stop_after=None, bank_rows=None, **unused):
        """sample [b,4,h,w] -> hidden state [b,320,h,w] (the reference discards it; the banks are the product)."""
        b, c, h, w = sample.shape
        dt = self.compute_dtype
        if self.precision == "split":
            x = torch.zeros((b, h, w, 8), device=sample.device, dtype=torch.float32)
            x[..., :c] = sample.float().permute(0, 2, 3, 1)
        else:
            x = ops.ncfhw_to_tokens(sample.contiguous()[:, :, None], dt, cpad=8)
        ctx = Ctx(dt, b, 1)
        ctx.stop_after, ctx.bank_rows = stop_after, bank_rows
        try:
            y = self.run_tokens(x, timestep, encoder_hidden_states, b, 1, None, ctx)
        except EarlyExit:
            return None
        out = y.permute(0, 3, 1, 2).contiguous().to(sample.dtype)
        return _Out(sample=out) if return_dict else (out,)

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, **kw):
        path = Path(pretrained_model_path)
        if subfolder is not None:
            path = path / subfolder
        cfg = json.loads((path / "config.json").read_text())
        names = ("sample_size", "in_channels", "block_out_channels", "layers_per_block", "norm_num_groups", "norm_eps",
                 "cross_attention_dim", "attention_head_dim")
        model = cls(**{k: cfg[k] for k in names if k in cfg})
        sd = {k: v for k, v in _load_weights(path).items() if not k.startswith(("conv_out.", "conv_norm_out."))}
        model.load_state_dict(sd, strict=False)
        return model


class _Out:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __getitem__(self, i):
        return list(self.__dict__.values())[i]


def _load_weights(path):
    st = path / "diffusion_pytorch_model.safetensors"
    if st.exists():
        from safetensors.torch import load_file
        return load_file(str(st), device="cpu")
    binf = path / "diffusion_pytorch_model.bin"
    if binf.exists():
        return torch.load(binf, map_location="cpu", weights_only=True)
    raise FileNotFoundError(f"no weights file found in {path}")


class ReferenceAttentionControl:
    """Drop-in for src/models/mutual_self_attention.py::ReferenceAttentionControl (fusion_blocks='full').

    write: every spatial block of the reference UNet banks norm1(x); read: the denoising UNet's blocks attend
    [self || bank] on cond rows.  update() pairs reader and writer blocks exactly like the reference (stable
    sort by channel count over the same traversal) and projects the bank to K/V ONCE (it is step- and
    frame-invariant), keeping it in the compute dtype (the reference casts to fp16, :313,349)."""

    def __init__(self, unet, mode="write", do_classifier_free_guidance=False, fusion_blocks="full", batch_size=1, **unused):
        assert mode in ("read", "write") and fusion_blocks == "full"
        self.unet, self.mode, self.cfg = unet, mode, do_classifier_free_guidance
        for blk in unet.spatial_blocks():
            blk.mode, blk.bank, blk.bank_kv = mode, [], None

    def last_block(self):
        """The transformer block executed last in forward order (after its bank write the writer graph is dead)."""
        return self.unet.up_blocks[-1].attentions[-1].transformer_blocks[0]

    def update(self, writer, dtype=None):
        for r, w in zip(self.unet.spatial_blocks(), writer.unet.spatial_blocks()):
            bank = w.bank[0]
            if self.unet.precision == "split":
                from . import precise
                precise.set_bank(r, bank[-1:], self.unet.compute_dtype)
            else:
                r.set_bank(bank[-1:], self.unet.compute_dtype)  # cond row: the only one cond queries ever read

    def clear(self):
        for blk in self.unet.spatial_blocks():
            blk.bank, blk.bank_kv = [], None
