"""Building blocks of the MI355X-native denoising path.

Each class is (a) a parameter container whose attribute paths reproduce the reference's
state-dict keys (so `denoising_unet.pth`, `reference_unet.pth`, `pose_guider.pth`,
`motion_module.pth`, the SD1.5 unet and sd-vae-ft-mse checkpoints load unchanged) and (b) a
`run()` method that executes the block with the HIP kernels of libmimo_hip.so through
mimo_amd.ops.  nn.Conv2d / nn.Linear / nn.GroupNorm / nn.LayerNorm are used purely as parameter
holders — their torch forward is never called; there is no torch compute fallback.

Data layout: activations are channels-last token-major [n, H, W, C] with
n = batch*frames images in frame-major order; the residual stream is fp32, every MFMA operand
is fp16/bf16; the reference's `rearrange "b c f h w <-> (b f) c h w"` copies do not exist.

Reference classes mirrored (paths relative to the reference repo):
  ResnetBlock             src/models/resnet.py:123-247 (ResnetBlock3D) / diffusers ResnetBlock2D
  Downsample, Upsample    src/models/resnet.py:31-120
  SpatialTransformerBlock src/models/attention.py:298-445 + patched forward src/models/mutual_self_attention.py:93-276
  SpatialTransformer      src/models/transformer_3d.py:27-169 (transformer_2d.py per image)
  MotionModule            src/models/motion_module.py:44-390
"""
import math

import torch
import torch.nn as nn

from . import ops
from .packing import pack_block_head_stream, pack_block_tail_stream, pack_conv, pack_ff2_kperm, pack_geglu, pack_proj_tail, pack_ln_fold


LOG2E = 1.4426950408889634


class Ctx:
    """Per-forward execution context."""

    def __init__(self, dtype, b, F):
        self.dtype = dtype      # MFMA operand dtype (torch.float16 | torch.bfloat16)
        self.b = b              # batch elements (CFG halves)
        self.F = F              # frames per batch element (1 for 2-D models)
        self.t_emb = None       # half [b, C0]: precomputed sinusoidal timestep embedding (UNetBase.timestep_table) or None
        self.temb = None        # fp32 [b, sum(Cout)]: every ResBlock's time_emb_proj(silu(emb)) at once
        self.attn2 = None       # fp32 [b, sum(C)]: every block's collapsed cross-attention output
        self.band_rows = None   # VAE tiled decode: images taller than this run GroupNorm-apply + conv per row band (exact halos)
        self.stop_after = None  # write mode: block after whose bank write the rest of the graph is dead
        self.bank_rows = None   # write mode: batch rows to bank (None = all)


class EarlyExit(Exception):
    pass


def row_bands(H, rows):
    """[(y0, y1)] covering [0, H) in bands of `rows` output rows (the last one may be shorter)."""
    return [(y0, min(H, y0 + rows)) for y0 in range(0, H, rows)]


def banded(ctx, H, W):
    """True when an H x W image is processed in row bands: tiling is on and the image is taller than a band.  Small
    images (the ones whose GroupNorm statistics ride in the producers' epilogues) always run whole."""
    return bool(ctx.band_rows) and H > ctx.band_rows and H * W > ops.COLSTATS_MAX_HW


def gn_conv3x3_banded(ctx, x, stats, gamma, beta, groups, w, cout, bias, out, *, raw_shortcut=False, residual=None,
                      out_scale=1.0):
    """out = conv3x3(silu(GroupNorm(x))) computed per (image, row band): the normalised half tensor exists only one band
    (+ one halo row above and below) at a time.  `stats` are the statistics of the WHOLE images, so every output pixel
    sees exactly the operands of the untiled launch, and every output element accumulates its K = 9 Cin products in the
    same order: the result is bit-identical to the untiled conv (split-K, the one batch-size dependent choice, is off
    for the band launches; the untiled launch of such a large image never splits either).
    raw_shortcut: `w` carries the fused 1x1 shortcut over half(x) as a 10th K segment (ResnetBlock)."""
    n, H, W, _ = x.shape
    with ops.split_k(False):
        for i in range(n):
            for y0, y1 in row_bands(H, ctx.band_rows):
                lo, hi = max(0, y0 - 1), min(H, y1 + 1)
                a, _ = ops.group_norm_apply(x[i:i + 1, lo:hi], stats[i:i + 1], gamma, beta, groups=groups, silu=True, dtype=ctx.dtype)
                raw = None
                if raw_shortcut:
                    _, raw = ops.group_norm_apply(x[i:i + 1, y0:y1], None, None, None, dtype=ctx.dtype, want_norm=False, want_raw=True)
                ops.conv2d(a, w, cout, pad=(1 if y0 == 0 else 0, 1), out_hw=(y1 - y0, W), x2=raw, bias=bias,
                           residual=None if residual is None else residual[i:i + 1, y0:y1], out_f32=True,
                           out_scale=out_scale, out=out[i:i + 1, y0:y1])
    return out


_PACK_EPOCH = [0]


def pack_epoch():
    """Generation counter of the packed-weight caches: bumped whenever any HipModule drops its packed buffers.  Captured
    hipGraphs hold those buffers by address and compare this before every replay (pipeline.GraphedDenoiser)."""
    return _PACK_EPOCH[0]


class HipModule(nn.Module):
    """Caches device-side packed weights per (dtype, device); invalidated by load_state_dict / _apply."""

    def packed(self, dtype):
        key = (dtype, self._dev())
        cache = self.__dict__.setdefault("_pk", {})
        if key not in cache:
            if cache:
                _PACK_EPOCH[0] += 1  # another (dtype, device) replaces the buffers a captured graph may point at
            cache.clear()
            with torch.no_grad():
                cache[key] = self._pack(dtype)
        return cache[key]

    def _dev(self):
        return next(self.parameters()).device

    def packed_split(self, dtype, fn):
        """A second packed-weight cache (same invalidation) for the split-operand precision policy of the VAE (vae.py):
        fn(dtype) -> dict, built on first use."""
        key = (dtype, self._dev())
        cache = self.__dict__.setdefault("_pk_split", {})
        if key not in cache:
            cache.clear()
            with torch.no_grad():
                cache[key] = fn(dtype)
        return cache[key]

    def prepack(self, dtype):
        """Pack every HipModule below (and including) this one NOW, on the current stream.  `packed()` packs lazily on
        whichever stream is current and publishes the cache entry at once; a consumer on ANOTHER stream would then read
        buffers whose pack kernels it is not ordered behind, and the buffers would live in that stream's allocator pool.
        The pipeline calls this on the main stream before it forks side streams (their wait_stream(main) orders them)."""
        for m in self.modules():
            if isinstance(m, HipModule) and hasattr(m, "_pack"):
                m.packed(dtype)
        return self

    def invalidate(self):
        _PACK_EPOCH[0] += 1
        for m in self.modules():
            m.__dict__.pop("_pk", None)
            m.__dict__.pop("_pk_split", None)
            m.__dict__.pop("_pk_split_parts", None)

    def _apply(self, fn, *a, **k):
        self.invalidate()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.invalidate()
        return super().load_state_dict(*a, **k)


def _f32(t):
    return t.detach().float().contiguous()


class ResnetBlock(HipModule):
    def __init__(self, in_channels, out_channels, temb_channels, groups=32, eps=1e-5, output_scale_factor=1.0):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels else None
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None
        self.in_channels, self.out_channels = in_channels, out_channels
        self.groups, self.eps, self.output_scale_factor = groups, eps, output_scale_factor
        self.temb_slice = None  # (start, end) columns of Ctx.temb, set by the owning UNet

    def _pack(self, dt):
        sc = self.conv_shortcut
        b2 = _f32(self.conv2.bias)
        if sc is not None:
            b2 = b2 + _f32(sc.bias)
        return dict(w1=pack_conv(self.conv1.weight, dt), b1=_f32(self.conv1.bias),
                    w2=pack_conv(self.conv2.weight, dt, shortcut=None if sc is None else sc.weight), b2=b2,
                    g1=_f32(self.norm1.weight), be1=_f32(self.norm1.bias),
                    g2=_f32(self.norm2.weight), be2=_f32(self.norm2.bias))

    def run(self, ctx, x, skip=None):
        """x: fp32 [n,H,W,C1]; skip: fp32 [n,H,W,C2] concatenated virtually on the channel axis.

        Each of the two norm -> SiLU -> conv3x3 stages is ONE launch where mimo_conv3x3_fused covers the layer (the large
        images: ops.hconv_supported, a function of the layer only): the convolution reads the fp32 tensor and applies the
        GroupNorm affine + SiLU on its LDS tile.  Elsewhere: GroupNorm-apply pass -> half tensor -> implicit-GEMM conv.
        `self.precision = "split"` (per module, opt-in): this block on split operands (mimo_amd.precise).
        `self.edge_parts` (set by the owning UNet on the resnets in front of its output head, ops.EDGE_SPLIT bit 3): those of the
        block's three products that take split operands under the DEFAULT policy — what they round reaches the prediction undamped."""
        if x.dtype == torch.float32:
            if getattr(self, "precision", "half") == "split":
                from . import precise
                return precise.resnet(self, ctx, x, skip)
            if getattr(self, "edge_parts", None) and (ops.EDGE_SPLIT & 8):
                from . import precise
                return precise.resnet(self, ctx, x, skip, parts=self.edge_parts)
        p = self.packed(ctx.dtype)
        fused_sc = self.conv_shortcut is not None
        n, H, W, _ = x.shape
        cout = self.out_channels
        band = skip is None and self.time_emb_proj is None and banded(ctx, H, W)
        tb = None
        if self.time_emb_proj is not None:
            s, e = self.temb_slice
            tb = ctx.temb[:, s:e]
        raw = None
        # ---- norm1 -> SiLU -> conv1 (+ time embedding) ----
        if ops.hconv_supported(x, cout, x2=skip):
            st1 = ops.group_norm_stats(x, groups=self.groups, eps=self.eps, x2=skip, dtype=ctx.dtype)
            ab1 = ops.group_norm_affine(st1, p["g1"], p["be1"], self.in_channels, self.groups)
            # the half cast of the raw input (operand of the fused shortcut in conv2) leaves as a side output
            r = ops.conv3x3_fused(x, p["w1"], cout, x2=skip, ab=ab1, bias=p["b1"], img_bias=tb, imgs_per_bias_row=ctx.F,
                                  want_raw=fused_sc and not band, raw_dtype=ctx.dtype, tile_stats=True)
            h, raw = r if (fused_sc and not band) else (r, None)
        elif band:
            st1 = ops.group_norm_stats(x, groups=self.groups, eps=self.eps, dtype=ctx.dtype)
            h = torch.empty((n, H, W, cout), device=x.device, dtype=torch.float32)
            gn_conv3x3_banded(ctx, x, st1, p["g1"], p["be1"], self.groups, p["w1"], cout, p["b1"], h)
        else:
            a1, raw = ops.group_norm(x, p["g1"], p["be1"], groups=self.groups, eps=self.eps, silu=True, x2=skip,
                                     dtype=ctx.dtype, want_raw=fused_sc)
            # colstats=True: the conv epilogue also emits the GroupNorm column statistics of its output, so the norm
            # that consumes it (norm2 here; the next block's norm for the block output) makes no statistics pass over HBM
            h = ops.conv2d(a1, p["w1"], cout, bias=p["b1"], img_bias=tb, imgs_per_bias_row=ctx.F, out_f32=True, colstats=True)
        # ---- norm2 -> SiLU -> conv2 (+ shortcut | residual) ----
        if not fused_sc and ops.hconv_supported(h, cout):
            st2 = ops.group_norm_stats(h, groups=self.groups, eps=self.eps, dtype=ctx.dtype)
            ab2 = ops.group_norm_affine(st2, p["g2"], p["be2"], cout, self.groups)
            return ops.conv3x3_fused(h, p["w2"], cout, ab=ab2, bias=p["b2"], residual=x, out_scale=1.0 / self.output_scale_factor,
                                     tile_stats=True)
        if band:
            st2 = ops.group_norm_stats(h, groups=self.groups, eps=self.eps, dtype=ctx.dtype)
            out = torch.empty_like(h)
            # conv2 reads GN2(h) per band; the fused shortcut segment reads half(x), the plain residual reads x
            with ops.split_k(False):
                for i in range(n):
                    for y0, y1 in row_bands(H, ctx.band_rows):
                        lo, hi = max(0, y0 - 1), min(H, y1 + 1)
                        a2, _ = ops.group_norm_apply(h[i:i + 1, lo:hi], st2[i:i + 1], p["g2"], p["be2"], groups=self.groups,
                                                     silu=True, dtype=ctx.dtype)
                        rawb = None
                        if fused_sc:
                            _, rawb = ops.group_norm_apply(x[i:i + 1, y0:y1], None, None, None, dtype=ctx.dtype, want_norm=False, want_raw=True)
                        ops.conv2d(a2, p["w2"], cout, pad=(1 if y0 == 0 else 0, 1), out_hw=(y1 - y0, W), x2=rawb,
                                   bias=p["b2"], residual=None if fused_sc else x[i:i + 1, y0:y1], out_f32=True,
                                   out_scale=1.0 / self.output_scale_factor, out=out[i:i + 1, y0:y1])
            return out
        a2, _ = ops.group_norm(h, p["g2"], p["be2"], groups=self.groups, eps=self.eps, silu=True, dtype=ctx.dtype)
        return ops.conv2d(a2, p["w2"], cout, x2=raw, bias=p["b2"],
                          residual=None if fused_sc else x, out_f32=True,
                          out_scale=1.0 / self.output_scale_factor, colstats=True)


class Downsample(HipModule):
    """3x3 stride-2 conv; padding=1 (UNets) or diffusers' asymmetric (0,1,0,1) pad when padding=0 (VAE encoder)."""

    def __init__(self, channels, out_channels=None, padding=1):
        super().__init__()
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, stride=2, padding=padding)
        self.padding = padding

    def _pack(self, dt):
        return dict(w=pack_conv(self.conv.weight, dt), b=_f32(self.conv.bias))

    def run(self, ctx, x):
        p = self.packed(ctx.dtype)
        _, xh = ops.group_norm(x, None, None, dtype=ctx.dtype, want_norm=False, want_raw=True)  # cast only
        n, H, W, _ = x.shape
        if self.padding == 0:
            return ops.conv2d(xh, p["w"], self.conv.out_channels, stride=2, pad=(0, 0),
                              out_hw=((H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1), bias=p["b"], out_f32=True, colstats=True)
        return ops.conv2d(xh, p["w"], self.conv.out_channels, stride=2, bias=p["b"], out_f32=True, colstats=True)


class Upsample(HipModule):
    """nearest x2 (or to an explicit size) folded into the 3x3 conv's gather."""

    def __init__(self, channels, out_channels=None):
        super().__init__()
        self.conv = nn.Conv2d(channels, out_channels or channels, 3, padding=1)

    def _pack(self, dt):
        return dict(w=pack_conv(self.conv.weight, dt), b=_f32(self.conv.bias))

    def run(self, ctx, x, output_size=None):
        p = self.packed(ctx.dtype)
        n, H, W, _ = x.shape
        if output_size is None and ops.hconv_supported(x, self.conv.out_channels, normed=False, upsample2x=True):
            # the convolution gathers the nearest-x2 image from the fp32 source itself: no cast pass, no half intermediate
            return ops.conv3x3_fused(x, p["w"], self.conv.out_channels, bias=p["b"], upsample2x=True, tile_stats=True)
        if output_size is None and banded(ctx, 2 * H, 2 * W):
            # row bands of the UPSAMPLED image (even boundaries): output rows [y0, y1) read virtual rows y0 - 1 .. y1, i.e.
            # source rows y0 / 2 - 1 .. y1 / 2; the band's virtual image starts at source row lo, so the first virtual row
            # the band needs is local row (y0 - 1) - 2 lo: pad_t = 2 lo + 1 - y0 (= 1 for the top band, -1 below it)
            out = torch.empty((n, 2 * H, 2 * W, self.conv.out_channels), device=x.device, dtype=torch.float32)
            R = max(2, ctx.band_rows // 2 * 2)
            with ops.split_k(False):
                for i in range(n):
                    for y0, y1 in row_bands(2 * H, R):
                        lo, hi = max(0, y0 // 2 - 1), min(H, y1 // 2 + 1)
                        _, xh = ops.group_norm(x[i:i + 1, lo:hi], None, None, dtype=ctx.dtype, want_norm=False, want_raw=True)
                        ops.conv2d(xh, p["w"], self.conv.out_channels, upsample_to=(2 * (hi - lo), 2 * W),
                                   pad=(2 * lo + 1 - y0, 1), out_hw=(y1 - y0, 2 * W), bias=p["b"], out_f32=True,
                                   out=out[i:i + 1, y0:y1])
            return out
        _, xh = ops.group_norm(x, None, None, dtype=ctx.dtype, want_norm=False, want_raw=True)
        size = (2 * H, 2 * W) if output_size is None else tuple(output_size)
        return ops.conv2d(xh, p["w"], self.conv.out_channels, upsample_to=size, bias=p["b"], out_f32=True, colstats=True)


class _Attn(nn.Module):
    """Parameter holder with the diffusers Attention key layout (to_q/to_k/to_v/to_out.0)."""

    def __init__(self, query_dim, cross_dim=None, heads=8, dim_head=64, bias=False):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(cross_dim or query_dim, inner, bias=bias)
        self.to_v = nn.Linear(cross_dim or query_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Identity()])


class _GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, 2 * inner)


class _FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([_GEGLU(dim, 4 * dim), nn.Identity(), nn.Linear(4 * dim, dim)])


def _fold_geglu(proj, norm, dt):
    """packing.pack_ln_fold of a GEGLU projection (GEGLU row packing first) with the LayerNorm in front of it."""
    w, b = pack_geglu(proj.weight, proj.bias, torch.float32)
    return pack_ln_fold(w, norm.weight, norm.bias, b, dt)


def _ff_fusable(p, n3, out_f32):
    # (batch-invariant runs — split-K off — take the fused kernel whatever the row count: a b = 1 unit and the b = 2 launch
    # of the same window must go through the same kernel, the two forms differ in their fp32 summation order)
    # (FF_FUSED_MAX_ROWS: the fused kernels address their operands with 32-bit byte offsets and return MIMO_EINVAL beyond
    # it — such a launch takes the 3-4 launch path instead of raising)
    return ops.FF_FUSED and not out_f32 and p.get("ff2_wk") is not None and n3.shape[0] <= ops.FF_FUSED_MAX_ROWS and \
        (n3.shape[0] >= ops.FF_FUSED_MIN_ROWS or not ops.split_k_enabled())


def _ff_run(ctx, p, x_f32, n3, out_f32):
    """(LN, already applied: n3) -> GEGLU GEMM -> GEMM + residual.  Returns fp32 (residual stream) or half (feeds a projection)."""
    if _ff_fusable(p, n3, out_f32):
        # C = 320: FF1 + GEGLU + FF2 + residual in ONE launch, the [M, 4C] intermediate stays on the chip (ff_fused.hip)
        return ops.ff_fused(n3, p["ff1_w"], p["ff1_b"], p["ff2_wk"], p["ff2_b"], x_f32)
    if isinstance(n3, ops.LnFold):  # C >= 640: the LayerNorm is folded into this projection (ops.LN_FOLD)
        f = p["ff1_f"]
        h = ops.gemm(n3, f["w"], bias=f["bias"], colsum=f["colsum"], geglu=True)
    else:
        h = ops.gemm(n3, p["ff1_w"], bias=p["ff1_b"], geglu=True)
    return ops.gemm(h, p["ff2_w"], bias=p["ff2_b"], residual=x_f32, out_f32=out_f32)


def _ff_proj_run(ctx, p, y_f32, n3, proj, x_f32, colstats):
    """Feed-forward + the block's output projection + its residual: x + proj_out(y + FF(n3)).  proj = dict(po_w, po_b,
    po_wk) of the owning module.  One launch where the fused kernel applies (C = 320), three otherwise."""
    if ops.FF_PROJ_FUSED and proj.get("po_wk") is not None and _ff_fusable(p, n3, False):
        return ops.ff_proj_fused(n3, p["ff1_w"], p["ff1_b"], p["ff2_wk"], p["ff2_b"], y_f32, proj["po_wk"], proj["po_b"], x_f32,
                                 colstats=colstats)
    z = _ff_run(ctx, p, y_f32, n3, out_f32=False)
    return ops.gemm(z, proj["po_w"], bias=proj["po_b"], residual=x_f32, out_f32=True, colstats=colstats)


def _head_fusable(ws, M, HW):
    """mimo_block_head_fused applies: C = 320 (a packed stream exists), at most two images / frames per 128-row panel, and the
    row-count rule of the fused tails (a function of the layer and the frame size, never of the batch beyond the threshold)."""
    return ops.BLOCK_HEAD_FUSED and ops.BLOCK_TAIL_FUSED and ops.FF_FUSED and ws is not None and HW >= 128 and \
        M <= ops.FF_FUSED_MAX_ROWS and (M >= ops.FF_FUSED_MIN_ROWS or not ops.split_k_enabled())


def _block_tail_run(ctx, p, o, t, proj, x_f32, keys, ln_eps, img_bias=None, rows_per_img=1, colstats=False):
    """Everything after a block's attention core + the owning transformer's proj_out in one launch (C = 320), or None when
    the fused kernel does not apply.  keys = (to_out bias, LN gamma, LN beta) names in p; proj["tail_ws"] = the weight stream."""
    ws = proj.get("tail_ws")
    if not (ops.BLOCK_TAIL_FUSED and ops.FF_PROJ_FUSED and ws is not None and _ff_fusable(p, o, False)):
        return None
    # (the kernel's per-image vector: 16-byte aligned rows, at most two images per 128-row panel; both conditions are
    # properties of the model and the frame size, not of the batch: a sharded unit and the full launch decide alike)
    if img_bias is not None and (img_bias.data_ptr() % 16 or img_bias.stride(0) % 4 or rows_per_img < 128):
        return None
    return ops.block_tail_fused(o, ws, p[keys[0]], t, p[keys[1]], p[keys[2]], ln_eps, p["ff1_b"], p["ff2_wk"], p["ff2_b"],
                                proj["po_b"], x_f32, img_bias=img_bias, rows_per_img=rows_per_img, colstats=colstats)


class SpatialTransformerBlock(HipModule):
    """mode None: plain; 'write': bank norm1(x) (reference UNet); 'read': cond rows attend [self || bank]."""

    def __init__(self, dim, heads, head_dim, cross_attention_dim):
        super().__init__()
        self.attn1 = _Attn(dim, None, heads, head_dim)
        self.norm1 = nn.LayerNorm(dim)
        self.attn2 = _Attn(dim, cross_attention_dim, heads, head_dim)
        self.norm2 = nn.LayerNorm(dim)
        self.ff = _FeedForward(dim)
        self.norm3 = nn.LayerNorm(dim)
        self.dim, self.heads = dim, heads
        self.mode = None
        self.bank = []        # write: [half [rows, N, C]]; read: same tensors handed over by update()
        self.bank_kv = None   # read: half [Nb, 2C] = bank . [W_k ; W_v]^T, computed once per clip
        self.attn2_slice = None

    def _pack(self, dt):
        a1 = self.attn1
        ff1_w, ff1_b = pack_geglu(self.ff.net[0].proj.weight, self.ff.net[0].proj.bias, dt)
        return dict(
            qkv=self.qkv_weight().to(dt).contiguous(),
            kv=torch.cat([a1.to_k.weight, a1.to_v.weight], 0).detach().to(dt).contiguous(),
            o_w=a1.to_out[0].weight.detach().to(dt).contiguous(), o_b=_f32(a1.to_out[0].bias),
            ff1_w=ff1_w, ff1_b=ff1_b, ff2_w=self.ff.net[2].weight.detach().to(dt).contiguous(),
            ff2_wk=pack_ff2_kperm(self.ff.net[2].weight, dt) if self.dim == ops.FF_FUSED_DIM else None,
            ff2_b=_f32(self.ff.net[2].bias),
            n1w=_f32(self.norm1.weight), n1b=_f32(self.norm1.bias),
            n3w=_f32(self.norm3.weight), n3b=_f32(self.norm3.bias),
            # C >= 640: norm1 folded into the QKV projection, norm3 into the GEGLU projection (ops.LN_FOLD).  Write mode (the
            # reference UNet) banks the NORMALISED tensor, so its norm1 is never folded: no folded QKV copy is packed for it
            qkv_f=pack_ln_fold(self.qkv_weight(), self.norm1.weight, self.norm1.bias, None, dt)
            if (self.dim >= ops.LN_FOLD_MIN_C and self.mode != "write") else None,
            ff1_f=_fold_geglu(self.ff.net[0].proj, self.norm3, dt) if self.dim >= ops.LN_FOLD_MIN_C else None)

    def qkv_weight(self):
        """fp32 [3C, C] = [W_q * softmax_scale * log2(e); W_k; W_v]: the attention kernel exponentiates the MFMA result as is."""
        a1 = self.attn1
        return torch.cat([a1.to_q.weight.detach().float() * ((self.dim // self.heads) ** -0.5 * LOG2E),
                          a1.to_k.weight.detach().float(), a1.to_v.weight.detach().float()], 0)

    def attn2_matrix(self):
        """attn2 over ONE key collapses exactly: softmax == 1 -> out = to_out(to_v(e)) independent of the query
        (src/models/attention.py:412-426 with encoder_hidden_states [b,1,768]).  Returns (W_o.W_v fp32 [C,768], b_o)."""
        a2 = self.attn2
        # weight preprocessing at pack time, like the repacking: folded once on the HOST in fp64 (no stock-library GEMM
        # on the device), rounded once to fp32
        wo, wv = a2.to_out[0].weight.detach().double().cpu(), a2.to_v.weight.detach().double().cpu()
        return (wo @ wv).float().to(a2.to_out[0].weight.device), _f32(a2.to_out[0].bias)

    def ln1(self, dtype):
        """norm1 as the `ln=` argument of the GEMM that produces this block's input (fused into its epilogue at C = 320,
        folded into the QKV projection at C >= 640 — except in write mode, which banks the normalised tensor itself)."""
        p = self.packed(dtype)
        # (fold only with a folded QKV copy at hand: a block packed in write mode and switched to another mode later runs unfolded)
        return dict(gamma=p["n1w"], beta=p["n1b"], eps=self.norm1.eps, fold=self.mode != "write" and p["qkv_f"] is not None)

    def run(self, ctx, t, n_img, N, out_f32=False, n1=None, proj=None, x=None, qkv=None):
        """t: fp32 tokens [n_img*N, C]; n1 = norm1(t) as half if the producer already computed it; qkv = the fused Q/K/V
        projection of norm1(t) if the producer computed that too (mimo_block_head_fused; not in write mode, which banks n1).
        Returns the block output (half unless out_f32) — or, with proj = the owning transformer's packed proj_out and
        x = its fp32 input tokens, the transformer's output x + proj_out(block output) (fp32)."""
        p = self.packed(ctx.dtype)
        C = self.dim
        if qkv is None:
            if n1 is None:
                n1 = ops.layer_norm(t, p["n1w"], p["n1b"], eps=self.norm1.eps, dtype=ctx.dtype)
            if self.mode == "write":
                bank = n1.view(n_img, N, C)
                self.bank.append(bank if ctx.bank_rows is None else bank[ctx.bank_rows])
                if ctx.stop_after is self:
                    raise EarlyExit()
            if isinstance(n1, ops.LnFold):
                f = p["qkv_f"]
                qkv = ops.gemm(n1, f["w"], bias=f["bias"], colsum=f["colsum"])
            else:
                qkv = ops.gemm(n1, p["qkv"])
        else:
            assert self.mode != "write"
        qkv = qkv.view(n_img, N, 3 * C)
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        if self.mode == "read" and self.bank_kv is not None:
            # rows [0, F) of a CFG batch are unconditional: self-attention only (mutual_self_attention.py:179-197)
            first = ctx.F if ctx.b == 2 else 0
            o = ops.attention(q, k, v, self.heads, k2=self.bank_kv[:, :C], v2=self.bank_kv[:, C:],
                              seg2_first_batch=first, q_prescaled=True)
        else:
            o = ops.attention(q, k, v, self.heads, q_prescaled=True)
        s, e = self.attn2_slice
        if proj is not None:
            out = _block_tail_run(ctx, p, o.view(-1, C), t, proj, x, ("o_b", "n3w", "n3b"), self.norm3.eps,
                                  img_bias=ctx.attn2[:, s:e], rows_per_img=ctx.F * N, colstats=N)
            if out is not None:
                return out
        # to_out + collapsed attn2 + residual, with norm3 of the result fused into the same epilogue (C = 320)
        y, n3 = ops.gemm(o.view(-1, C), p["o_w"], bias=p["o_b"], img_bias=ctx.attn2[:, s:e],
                         rows_per_img=ctx.F * N, residual=t, out_f32=True,
                         ln=dict(gamma=p["n3w"], beta=p["n3b"], eps=self.norm3.eps, fold=True))
        if proj is not None:
            return _ff_proj_run(ctx, p, y, n3, proj, x, N)
        return _ff_run(ctx, p, y, n3, out_f32)

    def set_bank(self, bank, dtype):
        """bank: half [1, Nb, C] (the cond reference features).  Projects K/V once (step- and frame-invariant)."""
        p = self.packed(dtype)
        self.bank = [bank]
        kv = ops.gemm(bank.reshape(-1, self.dim).to(dtype).contiguous(), p["kv"])
        # keep ONE persistent buffer per block: a captured hipGraph of the denoising forward reads it by address,
        # so a new clip refreshes its contents instead of re-capturing
        buf = self.__dict__.get("_bank_kv_buf")
        if buf is not None and buf.shape == kv.shape and buf.dtype == kv.dtype and buf.device == kv.device:
            buf.copy_(kv)
        else:
            self.__dict__["_bank_kv_buf"] = buf = kv
        self.bank_kv = buf


class SpatialTransformer(HipModule):
    def __init__(self, heads, head_dim, in_channels, cross_attention_dim, groups=32):
        super().__init__()
        inner = heads * head_dim
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6)
        self.proj_in = nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([SpatialTransformerBlock(inner, heads, head_dim, cross_attention_dim)])
        self.proj_out = nn.Conv2d(inner, in_channels, 1)
        self.groups = groups

    def _pack(self, dt):
        C = self.proj_in.out_channels
        blk = self.transformer_blocks[0]
        fusable = C == ops.FF_FUSED_DIM and self.proj_out.out_channels == C
        return dict(g=_f32(self.norm.weight), b=_f32(self.norm.bias),
                    tail_ws=pack_block_tail_stream(blk.attn1.to_out[0].weight,
                                                   pack_geglu(blk.ff.net[0].proj.weight, blk.ff.net[0].proj.bias, dt)[0],
                                                   self.proj_out.weight.detach().reshape(C, -1), dt) if fusable else None,
                    head_ws=pack_block_head_stream(self.proj_in.weight.detach().reshape(C, -1), blk.qkv_weight(), dt)
                    if fusable and self.proj_in.in_channels == C else None,
                    pi_w=self.proj_in.weight.detach().reshape(C, -1).to(dt).contiguous(), pi_b=_f32(self.proj_in.bias),
                    po_w=self.proj_out.weight.detach().reshape(self.proj_out.out_channels, -1).to(dt).contiguous(),
                    po_b=_f32(self.proj_out.bias),
                    po_wk=pack_proj_tail(self.proj_out.weight.detach().reshape(self.proj_out.out_channels, -1), dt)
                    if C == ops.FF_FUSED_DIM and self.proj_out.out_channels == C else None)

    def run(self, ctx, x):
        if getattr(self, "precision", "half") == "split" and x.dtype == torch.float32:
            from . import precise
            return precise.spatial_transformer(self, ctx, x)
        p = self.packed(ctx.dtype)
        n, H, W, C = x.shape
        blk = self.transformer_blocks[0]
        if blk.mode != "write" and x.dtype == torch.float32 and _head_fusable(p["head_ws"], n * H * W, H * W):
            # GroupNorm-apply + proj_in + norm1 + QKV in ONE launch (C = 320): the block is head, attention core, tail
            stats = ops.group_norm_stats(x, groups=self.groups, eps=1e-6, dtype=ctx.dtype)
            ab = ops.group_norm_affine(stats, p["g"], p["b"], C, groups=self.groups)
            bp = blk.packed(ctx.dtype)
            t, qkv = ops.block_head_fused(p["head_ws"], p["pi_b"], bp["n1w"], bp["n1b"], blk.norm1.eps, x=x.view(-1, C), gn_ab=ab,
                                          rows_per_img=H * W)
            out = blk.run(ctx, t, n, H * W, qkv=qkv, proj=p, x=x.view(-1, C))
            return ops.with_stats(out.view(n, H, W, C), ops.stats_of(out))
        g, _ = ops.group_norm(x, p["g"], p["b"], groups=self.groups, eps=1e-6, silu=False, dtype=ctx.dtype)
        t, n1 = ops.gemm(g.view(-1, C), p["pi_w"], bias=p["pi_b"], out_f32=True, ln=blk.ln1(ctx.dtype))
        out = blk.run(ctx, t, n, H * W, n1=n1, proj=p, x=x.view(-1, C))  # ... + proj_out + residual (one launch at C = 320)
        return ops.with_stats(out.view(n, H, W, C), ops.stats_of(out))


class _TemporalAttn(nn.Module):
    def __init__(self, dim, heads, max_len):
        super().__init__()
        self.heads = heads
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(dim, dim, bias=False)
        self.to_v = nn.Linear(dim, dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Identity()])
        self.pos_encoder = _PosEnc(dim, max_len)


class _PosEnc(nn.Module):  # src/models/motion_module.py:264-279 (persistent buffer => a state-dict key)
    def __init__(self, d_model, max_len):
        super().__init__()
        position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(1, max_len, d_model)
        pe[0, :, 0::2] = torch.sin(position * div_term)
        pe[0, :, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe)


class _TemporalBlock(nn.Module):
    def __init__(self, dim, heads, max_len):
        super().__init__()
        self.attention_blocks = nn.ModuleList([_TemporalAttn(dim, heads, max_len) for _ in range(2)])
        self.norms = nn.ModuleList([nn.LayerNorm(dim) for _ in range(2)])
        self.ff = _FeedForward(dim)
        self.ff_norm = nn.LayerNorm(dim)


class _TemporalTransformer(nn.Module):
    def __init__(self, in_channels, heads, max_len, groups=32):
        super().__init__()
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, in_channels)
        self.transformer_blocks = nn.ModuleList([_TemporalBlock(in_channels, heads, max_len)])
        self.proj_out = nn.Linear(in_channels, in_channels)


class MotionModule(HipModule):
    """VanillaTemporalModule: GN -> Linear -> 2x{LN(+PE) -> attention over frames -> +res} -> LN -> GEGLU FF -> Linear -> +res."""

    def __init__(self, in_channels, heads=8, max_len=32):
        super().__init__()
        self.temporal_transformer = _TemporalTransformer(in_channels, heads, max_len)
        nn.init.zeros_(self.temporal_transformer.proj_out.weight)  # zero_module, motion_module.py:72-75
        nn.init.zeros_(self.temporal_transformer.proj_out.bias)
        self.dim, self.heads, self.max_len = in_channels, heads, max_len

    def _pack(self, dt):
        tt = self.temporal_transformer
        blk = tt.transformer_blocks[0]
        h = lambda w: w.detach().to(dt).contiguous()
        ff1_w, ff1_b = pack_geglu(blk.ff.net[0].proj.weight, blk.ff.net[0].proj.bias, dt)
        d = dict(g=_f32(tt.norm.weight), b=_f32(tt.norm.bias), pi_w=h(tt.proj_in.weight), pi_b=_f32(tt.proj_in.bias),
                 po_w=h(tt.proj_out.weight), po_b=_f32(tt.proj_out.bias), ff1_w=ff1_w, ff1_b=ff1_b,
                 ff2_w=h(blk.ff.net[2].weight), ff2_b=_f32(blk.ff.net[2].bias),
                 ff2_wk=pack_ff2_kperm(blk.ff.net[2].weight, dt) if self.dim == ops.FF_FUSED_DIM else None,
                 po_wk=pack_proj_tail(tt.proj_out.weight, dt) if self.dim == ops.FF_FUSED_DIM else None,
                 tail_ws=pack_block_tail_stream(blk.attention_blocks[1].to_out[0].weight, ff1_w, tt.proj_out.weight, dt)
                 if self.dim == ops.FF_FUSED_DIM else None,
                 fnw=_f32(blk.ff_norm.weight), fnb=_f32(blk.ff_norm.bias))
        for i, (a, nrm) in enumerate(zip(blk.attention_blocks, blk.norms)):
            qkv_w = torch.cat([a.to_q.weight, a.to_k.weight, a.to_v.weight], 0).detach()
            # block head i: (GroupNorm + proj_in | the previous attention's to_out + residual) -> norms[i] + PE -> QKV
            lead = tt.proj_in.weight if i == 0 else blk.attention_blocks[i - 1].to_out[0].weight
            d[f"head_ws{i}"] = pack_block_head_stream(lead.detach(), qkv_w, dt) if self.dim == ops.FF_FUSED_DIM else None
            d[f"qkv{i}"] = torch.cat([a.to_q.weight, a.to_k.weight, a.to_v.weight], 0).detach().to(dt).contiguous()
            d[f"o_w{i}"], d[f"o_b{i}"] = h(a.to_out[0].weight), _f32(a.to_out[0].bias)
            d[f"nw{i}"], d[f"nb{i}"] = _f32(nrm.weight), _f32(nrm.bias)
            d[f"pe{i}"] = _f32(a.pos_encoder.pe[0])
            if self.dim >= ops.LN_FOLD_MIN_C:
                # norms[i] folded into this QKV projection (ops.LN_FOLD); the positional table, added BEHIND the LayerNorm
                # (motion_module.py:276-279), becomes a per-frame bias row pe[f] @ W^T of the projection
                d[f"qkv_f{i}"] = pack_ln_fold(qkv_w, nrm.weight, nrm.bias, None, dt)
                d[f"pew{i}"] = (a.pos_encoder.pe[0].detach().double() @ qkv_w.double().t()).float().contiguous()
        if self.dim >= ops.LN_FOLD_MIN_C:
            d["ff1_f"] = _fold_geglu(blk.ff.net[0].proj, blk.ff_norm, dt)
        return d

    def _pe_bias(self, p, i, b, F):
        """fp32 [b * F, 3C]: row img = the positional row of frame img % F through the QKV projection (cached per (i, b, F))."""
        cache = self.__dict__.setdefault("_pe_bias_cache", {})
        key = (i, b, F, p[f"pew{i}"].data_ptr())
        if key not in cache:
            if len(cache) >= 64:  # (b, F) pairs of a few clip shapes at most; entries of re-packed weights would pile up otherwise
                cache.clear()
                _PACK_EPOCH[0] += 1   # a captured hipGraph may hold a dropped entry by address: force its re-capture
            cache[key] = p[f"pew{i}"][:F].repeat(b, 1).contiguous()
        return cache[key]

    def run(self, ctx, x):
        if getattr(self, "precision", "half") == "split" and x.dtype == torch.float32:
            from . import precise
            return precise.motion_module(self, ctx, x)
        p = self.packed(ctx.dtype)
        n, H, W, C = x.shape
        HW = H * W
        if ctx.F > self.max_len:
            raise ValueError(f"window of {ctx.F} frames exceeds temporal_position_encoding_max_len={self.max_len}")
        blk_eps = self.temporal_transformer.transformer_blocks[0].ff_norm.eps
        if x.dtype == torch.float32 and _head_fusable(p["head_ws0"], n * HW, HW):
            # C = 320: five launches — head (GroupNorm + proj_in + LN + PE + QKV), attention over frames, head (to_out +
            # residual + LN + PE + QKV), attention, tail (to_out + residual + LN + feed-forward + proj_out + residual)
            norms = self.temporal_transformer.transformer_blocks[0].norms
            stats = ops.group_norm_stats(x, groups=32, eps=1e-6, dtype=ctx.dtype)
            ab = ops.group_norm_affine(stats, p["g"], p["b"], C, groups=32)
            t, qkv = ops.block_head_fused(p["head_ws0"], p["pi_b"], p["nw0"], p["nb0"], norms[0].eps, x=x.view(-1, C), gn_ab=ab,
                                          rows_per_img=HW, pe=p["pe0"], rows_per_frame=HW, pe_frames=ctx.F)
            o = ops.temporal_attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], ctx.b, ctx.F, HW, self.heads)
            t, qkv = ops.block_head_fused(p["head_ws1"], p["o_b0"], p["nw1"], p["nb1"], norms[1].eps, a=o, residual=t,
                                          pe=p["pe1"], rows_per_frame=HW, pe_frames=ctx.F)
            o = ops.temporal_attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], ctx.b, ctx.F, HW, self.heads)
            out = _block_tail_run(ctx, p, o, t, p, x.view(-1, C), ("o_b1", "fnw", "fnb"), blk_eps, colstats=HW)
            if out is None:
                t, u = ops.gemm(o, p["o_w1"], bias=p["o_b1"], residual=t, out_f32=True, ln=dict(gamma=p["fnw"], beta=p["fnb"]))
                out = _ff_proj_run(ctx, p, t, u, p, x.view(-1, C), H * W)
            return ops.with_stats(out.view(n, H, W, C), ops.stats_of(out))
        g, _ = ops.group_norm(x, p["g"], p["b"], groups=32, eps=1e-6, silu=False, dtype=ctx.dtype)
        # every LayerNorm (+ positional encoding) rides in the epilogue of the GEMM that produces its input
        if ops.ln_foldable(C, n * HW):  # C >= 640: each LayerNorm is folded into the projection that consumes it (ops.LN_FOLD)
            norms = self.temporal_transformer.transformer_blocks[0].norms
            ln = [dict(gamma=p[f"nw{i}"], beta=p[f"nb{i}"], eps=norms[i].eps, fold=True) for i in range(2)]
            ln.append(dict(gamma=p["fnw"], beta=p["fnb"], eps=blk_eps, fold=True))
        else:
            ln = [dict(gamma=p[f"nw{i}"], beta=p[f"nb{i}"], pe=p[f"pe{i}"], rows_per_frame=HW, pe_frames=ctx.F) for i in range(2)]
            ln.append(dict(gamma=p["fnw"], beta=p["fnb"]))
        t, u = ops.gemm(g.view(-1, C), p["pi_w"], bias=p["pi_b"], out_f32=True, ln=ln[0])
        out = None
        for i in range(2):
            if isinstance(u, ops.LnFold):
                f = p[f"qkv_f{i}"]
                qkv = ops.gemm(u, f["w"], bias=f["bias"], colsum=f["colsum"], img_bias=self._pe_bias(p, i, ctx.b, ctx.F),
                               rows_per_img=HW)
            else:
                qkv = ops.gemm(u, p[f"qkv{i}"])
            o = ops.temporal_attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], ctx.b, ctx.F, HW, self.heads)
            if i == 1:  # ... + to_out + residual + LayerNorm + feed-forward + proj_out + residual: one launch at C = 320
                out = _block_tail_run(ctx, p, o, t, p, x.view(-1, C), ("o_b1", "fnw", "fnb"), blk_eps, colstats=HW)
                if out is not None:
                    break
            t, u = ops.gemm(o, p[f"o_w{i}"], bias=p[f"o_b{i}"], residual=t, out_f32=True, ln=ln[i + 1])
        if out is None:
            out = _ff_proj_run(ctx, p, t, u, p, x.view(-1, C), H * W)
        return ops.with_stats(out.view(n, H, W, C), ops.stats_of(out))
