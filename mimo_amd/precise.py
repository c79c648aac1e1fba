"""The UNets under the split-operand precision policy (`unet.precision = "split"`, opt-in): every conv / Linear of
reference_unet and denoising_unet takes BOTH operands as hi + lo pairs of the 16-bit type — the activation pass (ops.split3:
GroupNorm + SiLU, or the fp32 LayerNorm output) writes [hi | hi | lo] channel blocks, the weight is packed [Whi | Wlo | Whi] along
K (packing.pack_*_split3) — through the SAME implicit-GEMM / GEMM kernels the fast path uses; operands that exist only as 16-bit
tensors (attention outputs, the GEGLU hidden state) meet a split WEIGHT in two accumulating launches.  Nothing here is a new matrix
kernel: it is the fast path's launches with 3x the K extent and none of the fusions (no fused head / tail / hconv / LayerNorm fold).

Why it exists: against the fp32 reference the default policy (16-bit operands, fp32 accumulation / statistics / residual stream)
sits at its own floor of 7e-4 on BASELINE configs[1] but at 1.1-1.4e-3 on configs[0] (four 250-step DDIM jumps) and on the
guidance-3.5 cases of the half-width test models — the fp16 rounding of the WEIGHTS alone is 8.7e-4 there
(profiles/r6_error_budget_config1_golden_inputs.txt).  Under this policy those cases meet the north star's 1e-3 (what remains is
the 16-bit Q / K / V, P and attention output: 3e-4), and bf16 — 8 mantissa bits, 6e-3 under the default policy — meets it too.
Cost: about 3x the MFMA work plus the unfused passes; a reference-grade mode, not the benchmarked one.

Reference ops covered: src/models/resnet.py:123-247, 31-120; attention.py:298-445 + mutual_self_attention.py:93-276;
transformer_3d.py:27-169; motion_module.py:44-390; unet_3d_edit_bkfill.py:447-520.
"""
import torch

from . import ops
from .modules import EarlyExit, LOG2E, _f32
from .packing import pack_conv_split3, pack_conv_taps, pack_geglu, pack_linear_split3, split_hi_lo


def _s3(x, dt, stats=None, g=None, b=None, groups=32, silu=False, c_off=0, c_total=0, ld=None):
    return ops.split3(x, stats, g, b, groups=groups, silu=silu, dtype=dt, c_off=c_off, c_total=c_total, ld=ld)


def _hi_lo(w, dt):
    hi, lo = split_hi_lo(w.reshape(w.shape[0], -1), dt)
    return hi.contiguous(), lo.contiguous()


def _gemm2(a, hi, lo, *, bias=None, img_bias=None, rows_per_img=0, residual=None):
    """fp32 [M, N] = residual + a @ (hi + lo)^T + bias for a 16-bit operand `a` and a split weight: two accumulating launches."""
    y = ops.gemm(a, hi, bias=bias, img_bias=img_bias, rows_per_img=rows_per_img, residual=residual, out_f32=True)
    return ops.gemm(a, lo, residual=y, out_f32=True)


def _conv_w_concat(weight, c1, dt, shortcut=None):
    """3x3 weight over a virtual concat [x (c1 channels) | skip]: per tap [W1hi W1lo W1hi | W2hi W2lo W2hi], matching
    torch.cat([split3(x), split3(skip)], -1); the fused 1x1 shortcut segment likewise."""
    co, ci = weight.shape[:2]
    if c1 >= ci:
        return pack_conv_split3(weight, dt, shortcut=shortcut)
    a = pack_conv_split3(weight[:, :c1], dt).reshape(co, 9, 3 * c1)
    b = pack_conv_split3(weight[:, c1:], dt).reshape(co, 9, 3 * (ci - c1))
    w = torch.cat([a, b], -1).reshape(co, -1)
    if shortcut is not None:
        s = shortcut.reshape(co, ci)
        w = torch.cat([w, pack_linear_split3(s[:, :c1], dt), pack_linear_split3(s[:, c1:], dt)], 1)
    return w.contiguous()


# ---- ResnetBlock / samplers ------------------------------------------------------------------------------------------
def resnet(blk, ctx, x, skip=None, parts=None):
    """ResnetBlock under split operands.  `parts` / `blk.split_parts` (default: all of "conv1", "conv2", "sc") selects WHICH of its
    three products take them — the 3x3 conv1, the 3x3 conv2, the fused 1x1 shortcut over the raw block input; the others run on
    plain 16-bit operands through the same launches (where a block's rounding matters: tools/sensitivity_scan.py,
    tools/config1_probe.py --scan; the default policy uses ("sc", "conv2") on the last resnets of the denoising UNet)."""
    dt = ctx.dtype
    c1 = x.shape[-1]
    sc = blk.conv_shortcut
    parts = frozenset(parts if parts is not None else getattr(blk, "split_parts", ("conv1", "conv2", "sc")))
    s1, s2, ss = "conv1" in parts, "conv2" in parts, ("sc" in parts or "sc_op" in parts)
    op_only = "sc_op" in parts   # probe: only the shortcut's OPERAND is split (its weight's low part zeroed)

    def pack(d):
        out = dict(w1=_conv_w_concat(blk.conv1.weight, c1, d) if s1 else None)
        w2 = pack_conv_split3(blk.conv2.weight, d) if s2 else blk.conv2.weight.detach().float().permute(0, 2, 3, 1).reshape(blk.out_channels, -1).to(d)
        if sc is not None:
            sw = sc.weight.reshape(sc.weight.shape[0], -1)
            if ss:
                segs = [pack_linear_split3(sw[:, :c1], d)] + ([pack_linear_split3(sw[:, c1:], d)] if c1 < sw.shape[1] else [])
                if op_only:
                    for sg in segs:
                        k = sg.shape[1] // 3
                        sg[:, k:2 * k] = 0
            else:
                segs = [sw.detach().to(d)]
            w2 = torch.cat([w2] + segs, 1)
        out["w2"] = w2.contiguous()
        return out
    cache = blk.__dict__.setdefault("_pk_split_parts", {})
    key = (parts, dt, c1)
    if key not in cache:
        with torch.no_grad():
            cache[key] = pack(dt)
    P = cache[key]
    p = blk.packed(dt)
    cout = blk.out_channels
    ct = blk.in_channels
    tb = None
    if blk.time_emb_proj is not None:
        s, e = blk.temb_slice
        tb = ctx.temb[:, s:e]
    st1 = ops.group_norm_stats(x, groups=blk.groups, eps=blk.eps, x2=skip, dtype=dt)
    raw16 = None
    if s1:
        if skip is None:
            a1 = _s3(x, dt, st1, p["g1"], p["be1"], blk.groups, True, 0, ct)
        else:   # both sources of the virtual concat side by side ([hi | hi | lo] each), no concat copy
            a1 = torch.empty(tuple(x.shape[:-1]) + (3 * ct,), device=x.device, dtype=dt)
            ops.split3(x, st1, p["g1"], p["be1"], groups=blk.groups, silu=True, dtype=dt, c_off=0, c_total=ct, out=a1, col=0)
            ops.split3(skip, st1, p["g1"], p["be1"], groups=blk.groups, silu=True, dtype=dt, c_off=c1, c_total=ct, out=a1, col=3 * c1)
        h = ops.conv2d(a1, P["w1"], cout, bias=p["b1"], img_bias=tb, imgs_per_bias_row=ctx.F, out_f32=True)
    elif ops.hconv_supported(x, cout, x2=skip):   # conv1 on 16-bit operands: the block's ordinary fused launch (GroupNorm + SiLU on the tile)
        ab1 = ops.group_norm_affine(st1, p["g1"], p["be1"], blk.in_channels, blk.groups)
        want_raw = sc is not None and not ss
        r = ops.conv3x3_fused(x, p["w1"], cout, x2=skip, ab=ab1, bias=p["b1"], img_bias=tb, imgs_per_bias_row=ctx.F,
                              want_raw=want_raw, raw_dtype=dt, tile_stats=True)
        h, raw16 = r if want_raw else (r, None)
    else:
        a1, raw16 = ops.group_norm_apply(x, st1, p["g1"], p["be1"], groups=blk.groups, silu=True, x2=skip, dtype=dt,
                                         want_raw=sc is not None and not ss)
        h = ops.conv2d(a1, p["w1"], cout, bias=p["b1"], img_bias=tb, imgs_per_bias_row=ctx.F, out_f32=True, colstats=True)
    st2 = ops.group_norm_stats(h, groups=blk.groups, eps=blk.eps, dtype=dt)
    if s2:
        a2 = _s3(h, dt, st2, p["g2"], p["be2"], blk.groups, True)
    else:
        a2, _ = ops.group_norm_apply(h, st2, p["g2"], p["be2"], groups=blk.groups, silu=True, dtype=dt)
    raw = None
    if sc is not None:
        if ss:
            if skip is None:
                raw = _s3(x, dt)
            else:
                raw = torch.empty(tuple(x.shape[:-1]) + (3 * ct,), device=x.device, dtype=dt)
                ops.split3(x, dtype=dt, out=raw, col=0)
                ops.split3(skip, dtype=dt, out=raw, col=3 * c1)
        else:
            if raw16 is None:
                _, raw16 = ops.group_norm_apply(x, None, None, None, x2=skip, dtype=dt, want_norm=False, want_raw=True)
            raw = raw16
    probe = getattr(blk, "split_probe", None)   # numerical probes (tools/config1_probe.py): "w_only" = the OPERANDS' low parts zeroed
    if probe == "w_only":                         # (= only the weights carry hi + lo), "op_only" = the weights' low parts zeroed
        if s2:
            a2[..., 2 * cout:] = 0
        if ss and raw is not None:
            if skip is None:
                raw[..., 2 * c1:] = 0
            else:
                raw[..., 2 * c1:3 * c1] = 0
                raw[..., 3 * c1 + 2 * (ct - c1):] = 0
    return ops.conv2d(a2, P["w2"], cout, x2=raw, bias=p["b2"], residual=None if sc is not None else x, out_f32=True,
                      out_scale=1.0 / blk.output_scale_factor, colstats=True)


def _conv_w_concat_conv2(blk, c1, dt):
    """conv2 (input: the block's own `cout` channels) + the fused 1x1 shortcut over the block INPUT (virtual concat split at c1)."""
    sc = blk.conv_shortcut
    w = pack_conv_split3(blk.conv2.weight, dt)
    if sc is None:
        return w
    s = sc.weight.reshape(sc.weight.shape[0], -1)
    segs = [pack_linear_split3(s[:, :c1], dt)] + ([pack_linear_split3(s[:, c1:], dt)] if c1 < s.shape[1] else [])
    return torch.cat([w] + segs, 1).contiguous()


def downsample(ds, ctx, x):
    P = ds.packed_split(ctx.dtype, lambda d: dict(w=pack_conv_split3(ds.conv.weight, d)))
    p = ds.packed(ctx.dtype)
    n, H, W, _ = x.shape
    xs = _s3(x, ctx.dtype)
    if ds.padding == 0:
        return ops.conv2d(xs, P["w"], ds.conv.out_channels, stride=2, pad=(0, 0), out_hw=((H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1),
                          bias=p["b"], out_f32=True)
    return ops.conv2d(xs, P["w"], ds.conv.out_channels, stride=2, bias=p["b"], out_f32=True)


def upsample(us, ctx, x, output_size=None):
    P = us.packed_split(ctx.dtype, lambda d: dict(w=pack_conv_split3(us.conv.weight, d)))
    p = us.packed(ctx.dtype)
    n, H, W, _ = x.shape
    size = (2 * H, 2 * W) if output_size is None else tuple(output_size)
    return ops.conv2d(_s3(x, ctx.dtype), P["w"], us.conv.out_channels, upsample_to=size, bias=p["b"], out_f32=True)


# ---- spatial transformer ---------------------------------------------------------------------------------------------
def _block_pack(blk, dt):
    a1 = blk.attn1
    ff1_w, ff1_b = pack_geglu(blk.ff.net[0].proj.weight, blk.ff.net[0].proj.bias, torch.float32)
    o_hi, o_lo = _hi_lo(a1.to_out[0].weight.detach().float(), dt)
    f_hi, f_lo = _hi_lo(blk.ff.net[2].weight.detach().float(), dt)
    kv_hi, kv_lo = _hi_lo(torch.cat([a1.to_k.weight, a1.to_v.weight], 0).detach().float(), dt)
    return dict(qkv=pack_linear_split3(blk.qkv_weight(), dt), o_hi=o_hi, o_lo=o_lo, ff1=pack_linear_split3(ff1_w, dt), ff1_b=ff1_b,
                ff2_hi=f_hi, ff2_lo=f_lo, kv_hi=kv_hi, kv_lo=kv_lo)


def transformer_block(blk, ctx, t, n_img, N):
    """SpatialTransformerBlock.run under the split policy: t fp32 [n_img * N, C] -> fp32 (block output, the residual stream)."""
    dt = ctx.dtype
    P = blk.packed_split(dt, lambda d: _block_pack(blk, d))
    p = blk.packed(dt)
    C = blk.dim
    n1 = ops.layer_norm(t, p["n1w"], p["n1b"], eps=blk.norm1.eps, out_f32=True)
    if blk.mode == "write":
        bank = n1.to(dt).view(n_img, N, C)   # the bank is a 16-bit tensor in the reference too (mutual_self_attention.py:313,349)
        blk.bank.append(bank if ctx.bank_rows is None else bank[ctx.bank_rows])
        if ctx.stop_after is blk:
            raise EarlyExit()
    qkv = ops.gemm(_s3(n1, dt), P["qkv"]).view(n_img, N, 3 * C)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    if blk.mode == "read" and blk.bank_kv is not None:
        first = ctx.F if ctx.b == 2 else 0
        o = ops.attention(q, k, v, blk.heads, k2=blk.bank_kv[:, :C], v2=blk.bank_kv[:, C:], seg2_first_batch=first, q_prescaled=True)
    else:
        o = ops.attention(q, k, v, blk.heads, q_prescaled=True)
    s, e = blk.attn2_slice
    y = _gemm2(o.view(-1, C), P["o_hi"], P["o_lo"], bias=p["o_b"], img_bias=ctx.attn2[:, s:e], rows_per_img=ctx.F * N, residual=t)
    n3 = ops.layer_norm(y, p["n3w"], p["n3b"], eps=blk.norm3.eps, out_f32=True)
    h = ops.gemm(_s3(n3, dt), P["ff1"], bias=P["ff1_b"], geglu=True)
    return _gemm2(h, P["ff2_hi"], P["ff2_lo"], bias=p["ff2_b"], residual=y)


def set_bank(blk, bank, dtype):
    """SpatialTransformerBlock.set_bank with the K / V projection's weight split (the bank itself is a 16-bit tensor)."""
    P = blk.packed_split(dtype, lambda d: _block_pack(blk, d))
    b2 = bank.reshape(-1, blk.dim).to(dtype).contiguous()
    kv32 = ops.gemm(b2, P["kv_hi"], out_f32=True)
    kv = ops.gemm(b2, P["kv_lo"], residual=kv32)
    blk.bank = [bank]
    buf = blk.__dict__.get("_bank_kv_buf")
    if buf is not None and buf.shape == kv.shape and buf.dtype == kv.dtype and buf.device == kv.device:
        buf.copy_(kv)
    else:
        blk.__dict__["_bank_kv_buf"] = buf = kv
    blk.bank_kv = buf


def spatial_transformer(tr, ctx, x):
    dt = ctx.dtype
    C = tr.proj_in.out_channels
    P = tr.packed_split(dt, lambda d: dict(pi=pack_linear_split3(tr.proj_in.weight.detach().reshape(C, -1), d),
                                           po=pack_linear_split3(tr.proj_out.weight.detach().reshape(tr.proj_out.out_channels, -1), d)))
    p = tr.packed(dt)
    n, H, W, Cx = x.shape
    st = ops.group_norm_stats(x, groups=tr.groups, eps=1e-6, dtype=dt)
    g3 = _s3(x, dt, st, p["g"], p["b"], tr.groups, False)
    t = ops.gemm(g3.view(-1, 3 * Cx), P["pi"], bias=p["pi_b"], out_f32=True)
    z = transformer_block(tr.transformer_blocks[0], ctx, t, n, H * W)
    out = ops.gemm(_s3(z, dt), P["po"], bias=p["po_b"], residual=x.view(-1, Cx), out_f32=True)
    return out.view(n, H, W, Cx)


# ---- motion module ---------------------------------------------------------------------------------------------------
def _motion_pack(mm, dt):
    tt = mm.temporal_transformer
    blk = tt.transformer_blocks[0]
    ff1_w, ff1_b = pack_geglu(blk.ff.net[0].proj.weight, blk.ff.net[0].proj.bias, torch.float32)
    d = dict(pi=pack_linear_split3(tt.proj_in.weight.detach(), dt), po=pack_linear_split3(tt.proj_out.weight.detach(), dt),
             ff1=pack_linear_split3(ff1_w, dt), ff1_b=ff1_b)
    d["ff2_hi"], d["ff2_lo"] = _hi_lo(blk.ff.net[2].weight.detach().float(), dt)
    for i, a in enumerate(blk.attention_blocks):
        d[f"qkv{i}"] = pack_linear_split3(torch.cat([a.to_q.weight, a.to_k.weight, a.to_v.weight], 0).detach(), dt)
        d[f"o_hi{i}"], d[f"o_lo{i}"] = _hi_lo(a.to_out[0].weight.detach().float(), dt)
    return d


def motion_module(mm, ctx, x):
    dt = ctx.dtype
    P = mm.packed_split(dt, lambda d: _motion_pack(mm, d))
    p = mm.packed(dt)
    n, H, W, C = x.shape
    HW = H * W
    if ctx.F > mm.max_len:
        raise ValueError(f"window of {ctx.F} frames exceeds temporal_position_encoding_max_len={mm.max_len}")
    blk = mm.temporal_transformer.transformer_blocks[0]
    st = ops.group_norm_stats(x, groups=32, eps=1e-6, dtype=dt)
    g3 = _s3(x, dt, st, p["g"], p["b"], 32, False)
    t = ops.gemm(g3.view(-1, 3 * C), P["pi"], bias=p["pi_b"], out_f32=True)
    for i in range(2):
        u = ops.layer_norm(t, p[f"nw{i}"], p[f"nb{i}"], eps=blk.norms[i].eps, pe=p[f"pe{i}"], rows_per_frame=HW, pe_frames=ctx.F,
                           out_f32=True)
        qkv = ops.gemm(_s3(u, dt), P[f"qkv{i}"])
        o = ops.temporal_attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], ctx.b, ctx.F, HW, mm.heads)
        t = _gemm2(o, P[f"o_hi{i}"], P[f"o_lo{i}"], bias=p[f"o_b{i}"], residual=t)
    u = ops.layer_norm(t, p["fnw"], p["fnb"], eps=blk.ff_norm.eps, out_f32=True)
    h = ops.gemm(_s3(u, dt), P["ff1"], bias=P["ff1_b"], geglu=True)
    z = _gemm2(h, P["ff2_hi"], P["ff2_lo"], bias=p["ff2_b"], residual=t)
    out = ops.gemm(_s3(z, dt), P["po"], bias=p["po_b"], residual=x.view(-1, C), out_f32=True)
    return out.view(n, H, W, C)


# ---- the UNet --------------------------------------------------------------------------------------------------------
def _unet_pack(u, dt):
    res = [m for m in u.modules() if type(m).__name__ == "ResnetBlock"]
    mats = [m.attn2_matrix() for m in u.spatial_blocks()]
    cin_pad = (u.in_channels + 7) // 8 * 8
    d = dict(ci=pack_conv_split3(u.conv_in.weight, dt, cin_pad=cin_pad, k_pad=(3 * cin_pad + 31) // 32 * 32 if cin_pad <= 8 else None),
             t1=pack_linear_split3(u.time_embedding.linear_1.weight.detach(), dt),
             t2=pack_linear_split3(u.time_embedding.linear_2.weight.detach(), dt),
             temb=pack_linear_split3(torch.cat([m.time_emb_proj.weight for m in res], 0).detach(), dt),
             a2=pack_linear_split3(torch.cat([w for w, _ in mats], 0), dt), cin_pad=cin_pad)
    if u.with_out:
        cp = (u.out_channels + 3) // 4 * 4
        d["co"] = pack_conv_split3(u.conv_out.weight, dt, cout_pad=cp)
        # the thin-output form (ops.conv3x3_thin_out): per-tap weight [9 cout, Cin] -> [9 cout, 3 Cin]
        d["co_t"] = pack_linear_split3(pack_conv_taps(u.conv_out.weight, torch.float32, cout_pad=cp), dt)
    return d


def clip_tables(u, timesteps, ehs, b):
    """UNetBase.clip_tables under the split policy: the sinusoid, both embedding layers and the CLIP embedding stay fp32 between
    the GEMMs (the default path rounds each of them to the 16-bit type)."""
    import math
    dt = u.compute_dtype
    P = u.packed_split(dt, lambda d: _unet_pack(u, d))
    p = u.packed(dt)
    S = len(timesteps)
    half = u.boc[0] // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    ang = torch.tensor([float(t) for t in timesteps], dtype=torch.float32)[:, None] * freqs[None, :]
    t_emb = torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)[:, None, :].repeat(1, b, 1).reshape(S * b, -1).contiguous().to(u.device)
    e1 = ops.gemm(_s3(t_emb, dt), P["t1"], bias=p["t1_b"], silu=True, out_f32=True)
    emb = ops.gemm(_s3(e1, dt), P["t2"], bias=p["t2_b"], silu=True, out_f32=True)
    temb = ops.gemm(_s3(emb, dt), P["temb"], bias=p["temb_b"], out_f32=True).view(S, b, -1)
    e = ehs.reshape(b, -1).to(device=u.device, dtype=torch.float32).contiguous()
    return temb, ops.gemm(_s3(e, dt), P["a2"], bias=p["a2_b"], out_f32=True)


def run_tokens(u, x_tok, timestep, ehs, b, F, pose_tok=None, ctx=None, temb=None, attn2=None):
    """UNetBase.run_tokens under the split policy.  x_tok: fp32 (or 16-bit: no low part then) [b*F, h, w, Cin_pad8]."""
    from .modules import Ctx
    dt = u.compute_dtype
    P = u.packed_split(dt, lambda d: _unet_pack(u, d))
    p = u.packed(dt)
    ctx = ctx or Ctx(dt, b, F)
    if temb is None or attn2 is None:
        tt, attn2 = clip_tables(u, [float(timestep)], ehs, b)
        temb = tt[0]
    ctx.temb, ctx.attn2 = temb, attn2
    n, H, W, _ = x_tok.shape
    up = 2 ** u.num_upsamplers
    forward_upsample_size = (H % up != 0) or (W % up != 0)
    x32 = x_tok if x_tok.dtype == torch.float32 else x_tok.float()
    ld = P["ci"].shape[1] // 9
    x = ops.conv2d(_s3(x32.contiguous(), dt, ld=ld), P["ci"], u.boc[0], bias=p["ci_b"],
                   residual=None if pose_tok is None else pose_tok.float(), out_f32=True)
    skips = [x]
    for blk in u.down_blocks:
        for i, res in enumerate(blk.resnets):
            x = resnet(res, ctx, x)
            if blk.has_attn:
                x = spatial_transformer(blk.attentions[i], ctx, x)
            if blk.has_motion:
                x = motion_module(blk.motion_modules[i], ctx, x)
            skips.append(x)
        if blk.downsamplers is not None:
            x = downsample(blk.downsamplers[0], ctx, x)
            skips.append(x)
    mid = u.mid_block
    x = resnet(mid.resnets[0], ctx, x)
    x = spatial_transformer(mid.attentions[0], ctx, x)
    if mid.has_motion:
        x = motion_module(mid.motion_modules[0], ctx, x)
    x = resnet(mid.resnets[1], ctx, x)
    for bi, blk in enumerate(u.up_blocks):
        nres = len(blk.resnets)
        res_skips, skips = skips[-nres:], skips[:-nres]
        size = None
        if bi != len(u.up_blocks) - 1 and forward_upsample_size:
            size = skips[-1].shape[1:3]
        for i, res in enumerate(blk.resnets):
            x = resnet(res, ctx, x, skip=res_skips.pop())
            if blk.has_attn:
                x = spatial_transformer(blk.attentions[i], ctx, x)
            if blk.has_motion:
                x = motion_module(blk.motion_modules[i], ctx, x)
        if blk.upsamplers is not None:
            x = upsample(blk.upsamplers[0], ctx, x, size)
    if not u.with_out:
        return x
    return output_head(u, x, P, p)


def output_head(u, x, P, p):
    """conv_norm_out + SiLU + conv_out with split operands: the thin-output form (tap GEMM + gather) where it applies."""
    dt = u.compute_dtype
    st = ops.group_norm_stats(x, groups=u.groups, eps=u.eps, dtype=dt)
    a = _s3(x, dt, st, p["no_g"], p["no_b"], u.groups, True)
    cp = P["co"].shape[0]
    if ops.THIN_OUT and cp <= 16:
        return ops.conv3x3_thin_out(a, P["co_t"], cp, bias=p["co_b"])
    return ops.conv2d(a, P["co"], cp, bias=p["co_b"], out_f32=True)
