"""sd-vae-ft-mse encode/decode and the pose guider on the HIP kernels.

  AutoencoderKL  drop-in for diffusers.AutoencoderKL as used by the reference
                 (run_animate.py:70-72; src/pipelines/pipeline_pose2vid_long_edit_bkfill_roiclip.py:71,120,430,438):
                 .encode(x).latent_dist.mean, .decode(z).sample, .config.block_out_channels, state-dict keys
                 of the diffusers checkpoint (new and deprecated attention key names accepted).
  PoseGuider     drop-in for src/models/pose_guider.py.

The reference encodes/decodes one frame per Python-loop iteration; here all frames go through each layer
as one batch (GroupNorm statistics are per image, so the arithmetic per frame is unchanged).
"""
import torch
import torch.nn as nn

from . import ops
from .modules import Ctx, Downsample, HipModule, ResnetBlock, Upsample, _f32, banded, gn_conv3x3_banded, row_bands
from .packing import pack_conv, pack_conv_split3, pack_conv_taps, pack_linear_split3, pad_vec, split_hi_lo


FLASH_HEAD_DIMS = (64, 80, 160, 512)  # single-head widths mimo_attention has a kernel for


# ---- split-operand precision policy ("split") -------------------------------------------------------------------------
# The 1e-3 parity bar is measured against the fp32 run of diffusers' AutoencoderKL (oracle); with 16-bit MFMA operands the
# ENCODER cannot meet it in isolation: rounding its weights to fp16 costs 1.27e-3 by itself, the 3x3-conv operands 1.13e-3
# (profiles/r4_error_budget_vae.txt), 1.15-1.18e-3 measured at 784x784.  Under this policy every conv / Linear of the VAE gets
# BOTH operands as hi + lo pairs: the activation pass (ops.split3: GroupNorm + SiLU or a plain cast) writes [hi | hi | lo]
# channel blocks, the weight is packed [Whi | Wlo | Whi] along K, and the ordinary implicit-GEMM kernel sums
# hi.Whi + hi.Wlo + lo.Whi in its fp32 accumulators — 3x the MFMA work of the layer, no new matrix kernel.  The fused
# convolution (hconv), column statistics and band tiling are not used here: the path is GroupNorm statistics -> split3 -> conv.
# Cost: the encoder is 2.3 % of a clip's FLOPs, and the animate path encodes ONE distinct background frame (pipeline dedup).


def _sp_resnet(blk, ctx, x):
    """ResnetBlock (no time embedding, no skip concat: the VAE's) under the split policy.  x fp32 [n, H, W, Cin] -> fp32."""
    sc = blk.conv_shortcut
    P = blk.packed_split(ctx.dtype, lambda dt: dict(
        w1=pack_conv_split3(blk.conv1.weight, dt),
        w2=pack_conv_split3(blk.conv2.weight, dt, shortcut=None if sc is None else sc.weight)))
    p = blk.packed(ctx.dtype)
    cout = blk.out_channels
    st1 = ops.group_norm_stats(x, groups=blk.groups, eps=blk.eps, dtype=ctx.dtype)
    a1 = ops.split3(x, st1, p["g1"], p["be1"], groups=blk.groups, silu=True, dtype=ctx.dtype)
    h = ops.conv2d(a1, P["w1"], cout, bias=p["b1"], out_f32=True)
    st2 = ops.group_norm_stats(h, groups=blk.groups, eps=blk.eps, dtype=ctx.dtype)
    a2 = ops.split3(h, st2, p["g2"], p["be2"], groups=blk.groups, silu=True, dtype=ctx.dtype)
    raw = ops.split3(x, dtype=ctx.dtype) if sc is not None else None
    return ops.conv2d(a2, P["w2"], cout, x2=raw, bias=p["b2"], residual=None if sc is not None else x, out_f32=True,
                      out_scale=1.0 / blk.output_scale_factor)


def _sp_downsample(ds, ctx, x):
    """diffusers' asymmetric (0, 1, 0, 1) pad + 3x3 stride-2 conv (VAE encoder) under the split policy."""
    P = ds.packed_split(ctx.dtype, lambda dt: dict(w=pack_conv_split3(ds.conv.weight, dt)))
    p = ds.packed(ctx.dtype)
    n, H, W, _ = x.shape
    assert ds.padding == 0
    return ops.conv2d(ops.split3(x, dtype=ctx.dtype), P["w"], ds.conv.out_channels, stride=2, pad=(0, 0),
                      out_hw=((H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1), bias=p["b"], out_f32=True)


def _sp_upsample(us, ctx, x):
    """nearest x2 folded into the 3x3 conv's gather (VAE decoder) under the split policy."""
    P = us.packed_split(ctx.dtype, lambda dt: dict(w=pack_conv_split3(us.conv.weight, dt)))
    p = us.packed(ctx.dtype)
    n, H, W, _ = x.shape
    return ops.conv2d(ops.split3(x, dtype=ctx.dtype), P["w"], us.conv.out_channels, upsample_to=(2 * H, 2 * W), bias=p["b"],
                      out_f32=True)


class _VaeAttn(nn.Module):
    def __init__(self, ch, groups, eps):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, ch, eps=eps)
        self.to_q = nn.Linear(ch, ch)
        self.to_k = nn.Linear(ch, ch)
        self.to_v = nn.Linear(ch, ch)
        self.to_out = nn.ModuleList([nn.Linear(ch, ch), nn.Identity()])


class VaeMidBlock(HipModule):
    """resnet -> single-head attention (d = C, GN eps 1e-6, residual) -> resnet  (diffusers UNetMidBlock2D)."""

    def __init__(self, ch, groups, eps):
        super().__init__()
        self.attentions = nn.ModuleList([_VaeAttn(ch, groups, eps)])
        self.resnets = nn.ModuleList([ResnetBlock(ch, ch, None, groups, eps), ResnetBlock(ch, ch, None, groups, eps)])
        self.ch, self.groups, self.eps = ch, groups, eps

    def _pack(self, dt):
        a = self.attentions[0]
        return dict(g=_f32(a.group_norm.weight), b=_f32(a.group_norm.bias),
                    qkv_w=torch.cat([a.to_q.weight, a.to_k.weight, a.to_v.weight], 0).detach().to(dt).contiguous(),
                    qkv_b=torch.cat([a.to_q.bias, a.to_k.bias, a.to_v.bias], 0).detach().float().contiguous(),
                    o_w=a.to_out[0].weight.detach().to(dt).contiguous(), o_b=_f32(a.to_out[0].bias))

    def run(self, ctx, x):
        p = self.packed(ctx.dtype)
        x = self.resnets[0].run(ctx, x)
        n, H, W, C = x.shape
        N = H * W
        g, _ = ops.group_norm(x, p["g"], p["b"], groups=self.groups, eps=self.eps, silu=False, dtype=ctx.dtype)
        qkv = ops.gemm(g.view(-1, C), p["qkv_w"], bias=p["qkv_b"]).view(n, N, 3 * C)
        if C in FLASH_HEAD_DIMS:
            # one head of d = C (512 for sd-vae-ft-mse): flash attention over the N tokens of every image in ONE launch
            # (attn512_kernel: the head dimension split over the four waves of a block) — no N x N scores in HBM
            o = ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], 1)
        else:
            o = self._attention_unfused(ctx, qkv, n, N, C)
        y = ops.gemm(o.view(-1, C), p["o_w"], bias=p["o_b"], residual=x.view(-1, C), out_f32=True).view(n, H, W, C)
        return self.resnets[1].run(ctx, y)


    def run_split(self, ctx, x):
        """The mid block under the split policy: the projections take split operands (the attention output, a 16-bit tensor,
        meets a split WEIGHT in two accumulating launches); the attention core itself stays on half Q / K / V — its rounding
        classes are 4e-6 .. 8e-5 on this model (profiles/r4_error_budget_vae.txt: QKV, P, ATT)."""
        a = self.attentions[0]
        P = self.packed_split(ctx.dtype, lambda dt: dict(
            qkv_w=pack_linear_split3(torch.cat([a.to_q.weight, a.to_k.weight, a.to_v.weight], 0), dt),
            o_hi=split_hi_lo(a.to_out[0].weight, dt)[0].contiguous(), o_lo=split_hi_lo(a.to_out[0].weight, dt)[1].contiguous()))
        p = self.packed(ctx.dtype)
        x = _sp_resnet(self.resnets[0], ctx, x)
        n, H, W, C = x.shape
        N = H * W
        st = ops.group_norm_stats(x, groups=self.groups, eps=self.eps, dtype=ctx.dtype)
        g3 = ops.split3(x, st, p["g"], p["b"], groups=self.groups, silu=False, dtype=ctx.dtype)
        qkv = ops.gemm(g3.view(-1, 3 * C), P["qkv_w"], bias=p["qkv_b"]).view(n, N, 3 * C)
        if C in FLASH_HEAD_DIMS:
            o = ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], 1)
        else:
            o = self._attention_unfused(ctx, qkv, n, N, C)
        y = ops.gemm(o.view(-1, C), P["o_hi"], bias=p["o_b"], residual=x.view(-1, C), out_f32=True)
        y = ops.gemm(o.view(-1, C), P["o_lo"], residual=y, out_f32=True).view(n, H, W, C)
        return _sp_resnet(self.resnets[1], ctx, y)

    def _attention_unfused(self, ctx, qkv, n, N, C):
        """Widths without a flash kernel (not reached by sd-vae-ft-mse): scores / softmax / P.V as GEMM + row softmax +
        GEMM per image, the N x N score matrix in HBM."""
        Np = (N + 7) // 8 * 8  # key count padded to the 16-byte operand granule; padded keys get probability 0
        o = torch.empty((n, N, C), device=qkv.device, dtype=ctx.dtype)
        kp = torch.zeros((Np, C), device=qkv.device, dtype=ctx.dtype)
        vt = torch.zeros((C, Np), device=qkv.device, dtype=ctx.dtype)
        pr = torch.zeros((N, Np), device=qkv.device, dtype=ctx.dtype)
        for i in range(n):
            kp[:N].copy_(qkv[i, :, C:2 * C])
            vt[:, :N].copy_(qkv[i, :, 2 * C:].t())
            s = ops.gemm(qkv[i, :, :C], kp, out_f32=True)
            ops.softmax_rows(s[:, :N], ctx.dtype, scale=C ** -0.5, out=pr[:, :N])
            ops.gemm(pr, vt, out=o[i])
        return o


class _EncDownBlock(nn.Module):
    def __init__(self, cin, cout, layers, add_down, groups, eps):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock(cin if i == 0 else cout, cout, None, groups, eps) for i in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample(cout, cout, padding=0)]) if add_down else None

    def run(self, ctx, x):
        for r in self.resnets:
            x = r.run(ctx, x)
        return x if self.downsamplers is None else self.downsamplers[0].run(ctx, x)


class _DecUpBlock(nn.Module):
    def __init__(self, cin, cout, layers, add_up, groups, eps):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock(cin if i == 0 else cout, cout, None, groups, eps) for i in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample(cout, cout)]) if add_up else None

    def run(self, ctx, x):
        for r in self.resnets:
            x = r.run(ctx, x)
        return x if self.upsamplers is None else self.upsamplers[0].run(ctx, x)


class Encoder(HipModule):
    def __init__(self, in_channels, out_channels, boc, layers, groups, eps=1e-6):
        super().__init__()
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        c = boc[0]
        for i, co in enumerate(boc):
            self.down_blocks.append(_EncDownBlock(c, co, layers, i != len(boc) - 1, groups, eps))
            c = co
        self.mid_block = VaeMidBlock(c, groups, eps)
        self.conv_norm_out = nn.GroupNorm(groups, c, eps=eps)
        self.conv_out = nn.Conv2d(c, 2 * out_channels, 3, padding=1)
        self.groups, self.eps = groups, eps

    def _pack(self, dt):
        return dict(ci_w=pack_conv(self.conv_in.weight, dt, cin_pad=8), ci_b=_f32(self.conv_in.bias),
                    g=_f32(self.conv_norm_out.weight), b=_f32(self.conv_norm_out.bias),
                    co_w=pack_conv(self.conv_out.weight, dt), co_b=_f32(self.conv_out.bias))

    def run(self, ctx, x_tok):
        p = self.packed(ctx.dtype)
        x = ops.conv2d(x_tok, p["ci_w"], self.conv_in.out_channels, bias=p["ci_b"], out_f32=True)
        for blk in self.down_blocks:
            x = blk.run(ctx, x)
        x = self.mid_block.run(ctx, x)
        a, _ = ops.group_norm(x, p["g"], p["b"], groups=self.groups, eps=self.eps, silu=True, dtype=ctx.dtype)
        return ops.conv2d(a, p["co_w"], self.conv_out.out_channels, bias=p["co_b"])  # half: feeds quant_conv


    def run_split(self, ctx, x32, co_w, co_b, cout):
        """Split policy: x32 fp32 [n, H, W, 8] (RGB + 5 zero channels); co_w / co_b: the split-packed output convolution the
        owner wants applied behind conv_norm_out (AutoencoderKL folds quant_conv's mean half into conv_out).  fp32 out."""
        P = self.packed_split(ctx.dtype, lambda dt: dict(ci_w=pack_conv_split3(self.conv_in.weight, dt, cin_pad=8, k_pad=32)))
        p = self.packed(ctx.dtype)
        x = ops.conv2d(ops.split3(x32, dtype=ctx.dtype, ld=32), P["ci_w"], self.conv_in.out_channels, bias=p["ci_b"], out_f32=True)
        for blk in self.down_blocks:
            for r in blk.resnets:
                x = _sp_resnet(r, ctx, x)
            if blk.downsamplers is not None:
                x = _sp_downsample(blk.downsamplers[0], ctx, x)
        x = self.mid_block.run_split(ctx, x)
        st = ops.group_norm_stats(x, groups=self.groups, eps=self.eps, dtype=ctx.dtype)
        a = ops.split3(x, st, p["g"], p["b"], groups=self.groups, silu=True, dtype=ctx.dtype)
        return ops.conv2d(a, co_w, cout, bias=co_b, out_f32=True)


class Decoder(HipModule):
    def __init__(self, in_channels, out_channels, boc, layers, groups, eps=1e-6):
        super().__init__()
        rev = list(reversed(boc))
        self.conv_in = nn.Conv2d(in_channels, rev[0], 3, padding=1)
        self.mid_block = VaeMidBlock(rev[0], groups, eps)
        self.up_blocks = nn.ModuleList()
        c = rev[0]
        for i, co in enumerate(rev):
            self.up_blocks.append(_DecUpBlock(c, co, layers + 1, i != len(rev) - 1, groups, eps))
            c = co
        self.conv_norm_out = nn.GroupNorm(groups, c, eps=eps)
        self.conv_out = nn.Conv2d(c, out_channels, 3, padding=1)
        self.groups, self.eps = groups, eps

    def _pack(self, dt):
        return dict(ci_w=pack_conv(self.conv_in.weight, dt, cin_pad=8), ci_b=_f32(self.conv_in.bias),
                    g=_f32(self.conv_norm_out.weight), b=_f32(self.conv_norm_out.bias),
                    co_w=pack_conv(self.conv_out.weight, dt, cout_pad=4), co_b=pad_vec(self.conv_out.bias, 4),
                    co_wt=pack_conv_taps(self.conv_out.weight, dt, cout_pad=4))

    def run(self, ctx, z_tok):
        p = self.packed(ctx.dtype)
        x = ops.conv2d(z_tok, p["ci_w"], self.conv_in.out_channels, bias=p["ci_b"], out_f32=True)
        x = self.mid_block.run(ctx, x)
        for blk in self.up_blocks:
            x = blk.run(ctx, x)
        if banded(ctx, x.shape[1], x.shape[2]):
            st = ops.group_norm_stats(x, groups=self.groups, eps=self.eps, dtype=ctx.dtype)
            out = torch.empty(tuple(x.shape[:3]) + (4,), device=x.device, dtype=torch.float32)
            if not ops.THIN_OUT:
                return gn_conv3x3_banded(ctx, x, st, p["g"], p["b"], self.groups, p["co_w"], 4, p["co_b"], out)
            # per (image, row band) with one halo row above and below: the tap GEMM is per pixel and the gather sums the nine
            # contributions in a fixed order, so the interior rows of a band carry the bits of the untiled launch (the band's
            # own edge rows see zero padding where the image continues: they are computed and dropped)
            n, H = x.shape[0], x.shape[1]
            for i in range(n):
                for y0, y1 in row_bands(H, ctx.band_rows):
                    lo, hi = max(0, y0 - 1), min(H, y1 + 1)
                    a, _ = ops.group_norm_apply(x[i:i + 1, lo:hi], st[i:i + 1], p["g"], p["b"], groups=self.groups, silu=True, dtype=ctx.dtype)
                    out[i:i + 1, y0:y1] = ops.conv3x3_thin_out(a, p["co_wt"], 4, bias=p["co_b"])[:, y0 - lo:y0 - lo + (y1 - y0)]
            return out
        a, _ = ops.group_norm(x, p["g"], p["b"], groups=self.groups, eps=self.eps, silu=True, dtype=ctx.dtype)
        if ops.THIN_OUT:
            return ops.conv3x3_thin_out(a, p["co_wt"], 4, bias=p["co_b"])  # [n,H,W,4], channel 3 is padding
        return ops.conv2d(a, p["co_w"], 4, bias=p["co_b"], out_f32=True)


def _decoder_run_split(dec, ctx, z32):
    """Decoder under the split policy (opt-in: AutoencoderKL.decode_precision = "split").  z32: fp32 [n, h, w, 8], the
    post_quant_conv output (channels 4..7 zero) -> fp32 [n, 8h, 8w, 4] like Decoder.run."""
    P = dec.packed_split(ctx.dtype, lambda dt: dict(ci_w=pack_conv_split3(dec.conv_in.weight, dt, cin_pad=8, k_pad=32),
                                                    co_w=pack_conv_split3(dec.conv_out.weight, dt, cout_pad=4)))
    p = dec.packed(ctx.dtype)
    x = ops.conv2d(ops.split3(z32, dtype=ctx.dtype, ld=32), P["ci_w"], dec.conv_in.out_channels, bias=p["ci_b"], out_f32=True)
    x = dec.mid_block.run_split(ctx, x)
    for blk in dec.up_blocks:
        for r in blk.resnets:
            x = _sp_resnet(r, ctx, x)
        if blk.upsamplers is not None:
            x = _sp_upsample(blk.upsamplers[0], ctx, x)
    st = ops.group_norm_stats(x, groups=dec.groups, eps=dec.eps, dtype=ctx.dtype)
    a = ops.split3(x, st, p["g"], p["b"], groups=dec.groups, silu=True, dtype=ctx.dtype)
    return ops.conv2d(a, P["co_w"], 4, bias=p["co_b"], out_f32=True)


def nchw_to_tokens32(x, cpad=8):
    """[n, C, H, W] (any float dtype, device) -> fp32 tokens [n, H, W, cpad] with zero padding channels: the input layout of
    the split-policy entry points (a layout copy of a 3 / 4-channel image: torch owns it, no arithmetic)."""
    n, C, H, W = x.shape
    out = torch.zeros((n, H, W, cpad), device=x.device, dtype=torch.float32)
    out[..., :C] = x.permute(0, 2, 3, 1)
    return out


class _Cfg(dict):
    __getattr__ = dict.__getitem__


class _Dist:
    def __init__(self, mean):
        self.mean = mean

    def mode(self):
        return self.mean


class _Out:
    def __init__(self, **kw):
        self.__dict__.update(kw)


_DEPRECATED_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


class AutoencoderKL(HipModule):
    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 latent_channels=4, norm_num_groups=32, act_fn="silu", sample_size=256, scaling_factor=0.18215, **unused):
        super().__init__()
        boc = list(block_out_channels)
        self.config = _Cfg(in_channels=in_channels, out_channels=out_channels, block_out_channels=boc,
                           layers_per_block=layers_per_block, latent_channels=latent_channels,
                           norm_num_groups=norm_num_groups, scaling_factor=scaling_factor)
        self.encoder = Encoder(in_channels, latent_channels, boc, layers_per_block, norm_num_groups)
        self.decoder = Decoder(latent_channels, out_channels, boc, layers_per_block, norm_num_groups)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        self.compute_dtype = torch.float16
        self.latent_channels = latent_channels
        self.tile_band_rows = None
        # Precision policy per direction: "split" = both operands of every conv / Linear as hi + lo pairs (3x the layer's
        # MFMA work, meets the 1e-3 bar in isolation), "half" = plain 16-bit operands (the fast path).  The encoder defaults
        # to "split": its isolated error under "half" is 1.15-1.18e-3 at the reference's 784x784; the decoder (4.2e-4 there) to "half".
        self.encode_precision = "split"
        self.decode_precision = "half"

    @property
    def dtype(self):
        return self.quant_conv.weight.dtype

    @property
    def device(self):
        return self.quant_conv.weight.device

    def enable_tiling(self, band_rows=64):
        """Tiled decode (BASELINE configs[4]: "VAE tiled decode"), result-identical to the untiled one: the reference has
        no blended tiling (only enable_vae_slicing, pipeline :82-86), so tiles must carry exact halos.  Every
        GroupNorm+SiLU -> 3x3 conv pair of the decoder levels taller than `band_rows` runs per row band with one halo
        row on each side (channels-last: a row band of an image is one contiguous slab), the nearest-x2 up-sampling
        convs likewise; GroupNorm statistics stay global per image (they have to: that is what makes the tiles exact).
        The normalised half intermediates then exist one band at a time."""
        self.tile_band_rows = int(band_rows)

    def disable_tiling(self):
        self.tile_band_rows = None

    def enable_slicing(self):  # the batched path already streams frames through each layer
        pass

    def disable_slicing(self):
        pass

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        if self.quant_conv.weight.dtype in (torch.float16, torch.bfloat16):
            self.compute_dtype = self.quant_conv.weight.dtype
        return r

    def load_state_dict(self, sd, strict=True, **kw):
        fixed = {}
        for k, v in sd.items():
            parts = k.split(".")
            if "attentions" in parts:
                for old, new in _DEPRECATED_ATTN.items():
                    if parts[-2] == old:
                        k = ".".join(parts[:-2] + [new, parts[-1]])
                        if v.dim() == 4:
                            v = v.reshape(v.shape[0], v.shape[1])
            fixed[k] = v
        return super().load_state_dict(fixed, strict=strict, **kw)

    @classmethod
    def from_pretrained(cls, path, **kw):
        import json
        from pathlib import Path
        from .unet import _load_weights
        path = Path(path)
        model = cls(**json.loads((path / "config.json").read_text()))
        model.load_state_dict(_load_weights(path))
        return model

    def _pack(self, dt):
        lc = self.latent_channels
        # quant_conv: only the mean half is consumed (latent_dist.mean); pad 4 -> 8 input ch of post_quant_conv
        return dict(q_w=self.quant_conv.weight.detach().reshape(2 * lc, 2 * lc)[:lc].to(dt).contiguous(),
                    q_b=_f32(self.quant_conv.bias)[:lc].contiguous(),
                    pq_w=torch.nn.functional.pad(self.post_quant_conv.weight.detach().float().reshape(lc, lc),
                                                 (0, 8 - lc, 0, 8 - lc)).to(dt).contiguous(),  # [8,8], zero pad rows/cols
                    pq_b=pad_vec(self.post_quant_conv.bias, 8))

    def max_images(self, H, W, split=False):
        """Images of H x W pixels one launch group may hold: the largest activation (block_out_channels[1] channels at
        full resolution, 2 bytes; three channel blocks under the split policy) must stay below the 2 GiB operand limit of
        mimo_conv2d (32-bit buffer offsets)."""
        boc = self.config.block_out_channels
        per_image = H * W * max(boc[0], boc[min(1, len(boc) - 1)]) * 2 * (3 if split else 1)
        return max(1, (2 ** 31 - 1) // per_image)

    def prepack(self, dtype):
        """HipModule.prepack + the split-policy packs of the directions that use them (built lazily otherwise: a side stream
        must not be the first to touch them)."""
        super().prepack(dtype)
        if "split" in (self.encode_precision, self.decode_precision):
            self.packed_split(dtype, self._split_pack)
        return self

    def _split_pack(self, dt):
        lc = self.latent_channels
        enc = self.encoder
        # conv_out followed by the mean half of quant_conv is ONE linear map: folded on the host in fp64 (no rounding of the
        # 8 moment channels in between: class MOM of the error budget, 2.2e-4), then split like every other weight
        wq = self.quant_conv.weight.detach().double().reshape(2 * lc, 2 * lc)[:lc].cpu()
        wco = enc.conv_out.weight.detach().double().cpu()
        w = torch.einsum("om,mikl->oikl", wq, wco).float().to(self.device)
        b = (wq @ enc.conv_out.bias.detach().double().cpu() + self.quant_conv.bias.detach().double().cpu()[:lc]).float().to(self.device)
        pq = torch.nn.functional.pad(self.post_quant_conv.weight.detach().float().reshape(lc, lc), (0, 8 - lc, 0, 8 - lc))
        return dict(coq_w=pack_conv_split3(w, dt, cout_pad=4), coq_b=pad_vec(b, 4), pq_w=pack_linear_split3(pq, dt))

    # ---- token-level API used by the pipeline ----
    def encode_tokens(self, x_tok):
        """x_tok: half [n,H,W,8] (RGB in [-1,1] + 5 zero channels) -> posterior mean, fp32 tokens [n,H/8,W/8,4]."""
        dt = self.compute_dtype
        ctx = Ctx(dt, x_tok.shape[0], 1)
        p = self.packed(dt)
        h = self.encoder.run(ctx, x_tok)
        n, hh, ww, c = h.shape
        return ops.gemm(h.view(-1, c), p["q_w"], bias=p["q_b"], out_f32=True).view(n, hh, ww, self.latent_channels)

    def encode_tokens_split(self, x32):
        """Split policy: x32 fp32 [n, H, W, 8] (RGB in [-1, 1] + 5 zero channels) -> posterior mean, fp32 tokens [n, H/8, W/8, 4]."""
        dt = self.compute_dtype
        ctx = Ctx(dt, x32.shape[0], 1)
        P = self.packed_split(dt, self._split_pack)
        lc = self.latent_channels
        cp = (lc + 3) // 4 * 4
        h = self.encoder.run_split(ctx, x32, P["coq_w"], P["coq_b"], cp)
        return h if cp == lc else h[..., :lc].contiguous()

    def decode_tokens_split(self, z32):
        """Split policy: z32 fp32 [n, h, w, 8] (4 latent channels + 4 zero) -> fp32 tokens [n, 8h, 8w, 4]."""
        dt = self.compute_dtype
        ctx = Ctx(dt, z32.shape[0], 1)
        P = self.packed_split(dt, self._split_pack)
        p = self.packed(dt)
        n, h, w, c = z32.shape
        z8 = ops.gemm(ops.split3(z32.view(-1, c), dtype=dt), P["pq_w"], bias=p["pq_b"], out_f32=True).view(n, h, w, 8)
        return _decoder_run_split(self.decoder, ctx, z8)

    def decode_tokens(self, z_tok):
        """z_tok: half [n,h,w,8] (4 latent channels + 4 zero) -> fp32 tokens [n,8h,8w,4] (RGB + 1 pad channel)."""
        dt = self.compute_dtype
        ctx = Ctx(dt, z_tok.shape[0], 1)
        ctx.band_rows = self.tile_band_rows
        p = self.packed(dt)
        n, h, w, c = z_tok.shape
        z8 = ops.gemm(z_tok.view(-1, c), p["pq_w"], bias=p["pq_b"]).view(n, h, w, 8)  # channels 4..7 stay zero
        return self.decoder.run(ctx, z8)

    # ---- diffusers-compatible surface ----
    def encode(self, x, return_dict=True):
        """x: [n,3,H,W] in [-1,1] -> .latent_dist.mean [n,4,H/8,W/8]"""
        if self.encode_precision == "split":
            m = self.encode_tokens_split(nchw_to_tokens32(x))
        else:
            tok = ops.ncfhw_to_tokens(x.contiguous()[:, :, None], self.compute_dtype, cpad=8)
            m = self.encode_tokens(tok)
        return _Out(latent_dist=_Dist(m.permute(0, 3, 1, 2).contiguous().to(x.dtype)))

    def decode(self, z, return_dict=True):
        """z: [n,4,h,w] -> .sample [n,3,8h,8w]"""
        if self.decode_precision == "split":
            y = self.decode_tokens_split(nchw_to_tokens32(z))
        else:
            tok = ops.ncfhw_to_tokens(z.contiguous()[:, :, None], self.compute_dtype, cpad=8)
            y = self.decode_tokens(tok)
        return _Out(sample=y[..., :3].permute(0, 3, 1, 2).contiguous().to(z.dtype))


class PoseGuider(HipModule):
    """conv 3->16, {16->16, 16->32 s2, 32->32, 32->96 s2, 96->96, 96->256 s2}, 256->C; SiLU between (pose_guider.py:47-57)."""

    def __init__(self, conditioning_embedding_channels=320, conditioning_channels=3, block_out_channels=(16, 32, 96, 256)):
        super().__init__()
        boc = block_out_channels
        self.conv_in = nn.Conv2d(conditioning_channels, boc[0], 3, padding=1)
        self.blocks = nn.ModuleList()
        for i in range(len(boc) - 1):
            self.blocks.append(nn.Conv2d(boc[i], boc[i], 3, padding=1))
            self.blocks.append(nn.Conv2d(boc[i], boc[i + 1], 3, padding=1, stride=2))
        self.conv_out = nn.Conv2d(boc[-1], conditioning_embedding_channels, 3, padding=1)
        nn.init.zeros_(self.conv_out.weight)  # zero_module (pose_guider.py:38-45)
        nn.init.zeros_(self.conv_out.bias)
        self.compute_dtype = torch.float16

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        if self.conv_in.weight.dtype in (torch.float16, torch.bfloat16):
            self.compute_dtype = self.conv_in.weight.dtype
        return r

    def _pack(self, dt):
        convs = [self.conv_in] + list(self.blocks) + [self.conv_out]
        return dict(w=[pack_conv(c.weight, dt, cin_pad=8 if i == 0 else None) for i, c in enumerate(convs)],
                    b=[_f32(c.bias) for c in convs])

    def run_tokens_split(self, x32):
        """Split policy (`precision = "split"`): x32 fp32 [F,H,W,8] -> fp32 tokens [F,H/8,W/8,C]; every conv with hi + lo operands."""
        dt = self.compute_dtype
        convs = [self.conv_in] + list(self.blocks) + [self.conv_out]

        def pack(d):
            out = []
            for i, c in enumerate(convs):
                cin = 8 if i == 0 else c.in_channels
                kp = (3 * cin + 31) // 32 * 32 if 3 * cin <= 32 else None
                out.append(pack_conv_split3(c.weight, d, cin_pad=cin, k_pad=kp))
            return dict(w=out)
        P = self.packed_split(dt, pack)
        p = self.packed(dt)
        x = x32
        for i, c in enumerate(convs):
            last = i == len(convs) - 1
            a = ops.split3(x.contiguous(), dtype=dt, ld=P["w"][i].shape[1] // 9)
            x = ops.conv2d(a, P["w"][i], c.out_channels, stride=c.stride[0], bias=p["b"][i], silu=not last, out_f32=True)
        return x

    def run_tokens(self, x_tok):
        """x_tok: half [F,H,W,8] (RGB in [0,1] + zero pad) -> fp32 tokens [F,H/8,W/8,C]."""
        if getattr(self, "precision", "half") == "split":
            return self.run_tokens_split(x_tok.float() if x_tok.dtype != torch.float32 else x_tok)
        p = self.packed(self.compute_dtype)
        convs = [self.conv_in] + list(self.blocks) + [self.conv_out]
        x = x_tok
        for i, c in enumerate(convs):
            last = i == len(convs) - 1
            x = ops.conv2d(x, p["w"][i], c.out_channels, stride=c.stride[0], bias=p["b"][i], silu=not last, out_f32=last)
        return x

    def forward(self, conditioning):
        """[1,3,F,H,W] -> [1,C,F,H/8,W/8] like the reference."""
        b, c, f, H, W = conditioning.shape
        tok = ops.ncfhw_to_tokens(conditioning.contiguous(), self.compute_dtype, cpad=8)
        y = self.run_tokens(tok)
        return ops.tokens_to_ncfhw(y, b, y.shape[-1], f, y.shape[1], y.shape[2]).to(conditioning.dtype)
