"""Block- and model-level parity on the GPU: product (HIP kernels through the C-ABI) vs the CPU fp32 oracle on
identical seeded weights and inputs.  Metric: relative L2 of the output.  Policy under test: fp16/bf16
only as MFMA inputs, fp32 accumulation / statistics / softmax / residual stream."""
import os

import pytest
import torch

from conftest import north_star, rel_l2
from helpers import MM4, build_pair_pose, build_pair_unets, build_pair_vae, small_kw

pytestmark = pytest.mark.gpu
DTYPES = [torch.float16, torch.bfloat16]
# single-module tolerances; whole-model numbers are logged and asserted separately below
TOL = {torch.float16: 1.5e-3, torch.bfloat16: 1.2e-2}
CFG_CAUSE = "CFG amplification u + 3.5 (c - u) on random-weight half-width models (profiles/r4_edge_case_bisect.txt)"
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report.txt")


def report(line):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(line + "\n")
    print(line)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def to_tok(x5):  # [b,c,f,h,w] -> [b*f,h,w,c]
    b, c, f, h, w = x5.shape
    return x5.permute(0, 2, 3, 4, 1).reshape(b * f, h, w, c).contiguous()


def from_tok(t, b, f):  # [b*f,h,w,c] -> [b,c,f,h,w]
    n, h, w, c = t.shape
    return t.reshape(b, f, h, w, c).permute(0, 4, 1, 2, 3).contiguous()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cskip,cout", [(320, 0, 320), (320, 0, 640), (640, 320, 320), (1280, 640, 640)])
def test_resnet_block(dev, dtype, cin, cskip, cout):
    from mimo_amd.modules import Ctx, ResnetBlock
    from oracle import models as OM, synth
    b, F, h = 2, 3, 8
    o = synth.build(OM.ResnetBlock3D, 3, in_channels=cin + cskip, out_channels=cout, temb_channels=1280)
    p = ResnetBlock(cin + cskip, cout, 1280)
    p.load_state_dict(o.state_dict(), strict=True)
    p.to(dev)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(b, cin, F, h, h, generator=g)
    skip = torch.randn(b, cskip, F, h, h, generator=g) if cskip else None
    temb = torch.randn(b, 1280, generator=g)
    ref = o(torch.cat([x, skip], 1) if cskip else x, temb)
    ctx = Ctx(dtype, b, F)
    ctx.temb = (torch.nn.functional.silu(temb) @ o.time_emb_proj.weight.t() + o.time_emb_proj.bias).to(dev).contiguous()
    p.temb_slice = (0, cout)
    out = p.run(ctx, to_tok(x).to(dev), None if skip is None else to_tok(skip).to(dev))
    e = rel_l2(from_tok(out.cpu(), b, F), ref)
    report(f"resnet {cin}+{cskip}->{cout} {dtype}: rel_l2={e:.2e}")
    assert e < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,heads,hw", [(320, 8, 8), (640, 8, 6), (1280, 8, 4)])
def test_spatial_transformer_read_mode(dev, dtype, C, heads, hw):
    """cond rows attend [self || bank], uncond rows self only, 1-key cross-attention collapsed to a bias."""
    from mimo_amd.modules import Ctx, SpatialTransformer
    from oracle import models as OM, synth
    b, F = 2, 3
    o = synth.build(OM.Transformer3DModel, 5, heads=heads, head_dim=C // heads, in_channels=C, cross_attention_dim=768)
    p = SpatialTransformer(heads, C // heads, C, 768)
    p.load_state_dict(o.state_dict(), strict=True)
    p.to(dev)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(b, C, F, hw, hw, generator=g)
    ehs = torch.cat([torch.zeros(1, 1, 768), torch.randn(1, 1, 768, generator=g)])
    bank = torch.randn(2, hw * hw, C, generator=g).to(torch.float16)  # reference casts banks to fp16
    ob = o.transformer_blocks[0]
    ob.mode, ob.bank, ob.do_cfg = "read", [bank], True
    ref = o(x, ehs)
    pb = p.transformer_blocks[0]
    pb.mode = "read"
    pb.attn2_slice = (0, C)
    pb.set_bank(bank[1:].to(dev), dtype)
    ctx = Ctx(dtype, b, F)
    w, bias = pb.attn2_matrix()
    ctx.attn2 = (ehs[:, 0].to(dev) @ w.t() + bias).contiguous()
    out = p.run(ctx, to_tok(x).to(dev))
    e = rel_l2(from_tok(out.cpu(), b, F), ref)
    report(f"spatial transformer C{C} {dtype}: rel_l2={e:.2e}")
    assert e < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,F,hw", [(320, 24, 4), (640, 8, 5), (1280, 24, 2)])
def test_motion_module(dev, dtype, C, F, hw):
    from mimo_amd.modules import Ctx, MotionModule
    from oracle import models as OM, synth
    b = 2
    o = synth.build(OM.VanillaTemporalModule, 7, in_channels=C)
    p = MotionModule(C)
    p.load_state_dict(o.state_dict(), strict=True)
    p.to(dev)
    x = torch.randn(b, C, F, hw, hw, generator=torch.Generator().manual_seed(3))
    ref = o(x)
    out = p.run(Ctx(dtype, b, F), to_tok(x).to(dev))
    e = rel_l2(from_tok(out.cpu(), b, F) - x, ref - x)  # compare the module's residual branch, not the identity
    report(f"motion module C{C} F{F} {dtype}: rel_l2(branch)={e:.2e}")
    assert e < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_level0_transformers_head_and_tail_fused_vs_oracle_and_unfused(dev, dtype):
    """C = 320 at 16 x 16 (256 rows per image = two 128-row panels): the spatial transformer as head (GroupNorm-apply + proj_in +
    norm1 + QKV) -> attention core -> tail, the motion module as head -> attention -> head (to_out + residual + LN + PE + QKV) ->
    attention -> tail — forced onto the fused kernels (split-K off lifts their row threshold, as in the sharded mode) — against
    the fp32 oracle AND against the same modules with BLOCK_HEAD_FUSED off (the launches the heads replace)."""
    from mimo_amd import ops
    from mimo_amd.modules import Ctx, MotionModule, SpatialTransformer
    from oracle import models as OM, synth
    C, heads, hw, b, F = 320, 8, 16, 2, 3
    g = torch.Generator().manual_seed(2)
    x = torch.randn(b, C, F, hw, hw, generator=g)
    # spatial transformer, read mode
    o = synth.build(OM.Transformer3DModel, 5, heads=heads, head_dim=C // heads, in_channels=C, cross_attention_dim=768)
    p = SpatialTransformer(heads, C // heads, C, 768)
    p.load_state_dict(o.state_dict(), strict=True)
    p.to(dev)
    ehs = torch.cat([torch.zeros(1, 1, 768), torch.randn(1, 1, 768, generator=g)])
    bank = torch.randn(2, hw * hw, C, generator=g).to(torch.float16)
    ob = o.transformer_blocks[0]
    ob.mode, ob.bank, ob.do_cfg = "read", [bank], True
    ref = o(x, ehs)
    pb = p.transformer_blocks[0]
    pb.mode = "read"
    pb.attn2_slice = (0, C)
    pb.set_bank(bank[1:].to(dev), dtype)
    ctx = Ctx(dtype, b, F)
    w, bias = pb.attn2_matrix()
    ctx.attn2 = (ehs[:, 0].to(dev) @ w.t() + bias).contiguous()
    xt = to_tok(x).to(dev)
    with ops.split_k(False):
        ops.COUNTER = {"flops": 0, "launches": 0}
        out = p.run(ctx, xt)
        n_fused = ops.COUNTER["launches"]
        ops.BLOCK_HEAD_FUSED = False
        try:
            ops.COUNTER = {"flops": 0, "launches": 0}
            out_u = p.run(ctx, xt)
            n_unfused = ops.COUNTER["launches"]
        finally:
            ops.BLOCK_HEAD_FUSED = True
            ops.COUNTER = None
    assert n_fused < n_unfused, (n_fused, n_unfused)   # the head really ran (GroupNorm-apply, GEMM + LN and QKV GEMM became one)
    e, eu = rel_l2(from_tok(out.cpu(), b, F), ref), rel_l2(out, out_u)
    report(f"spatial transformer C320 16x16 head + tail fused {dtype}: rel_l2={e:.2e} (vs unfused head {eu:.2e}; {n_fused} vs {n_unfused} launches)")
    assert e < TOL[dtype] and eu < TOL[dtype] / 2
    # motion module
    om = synth.build(OM.VanillaTemporalModule, 7, in_channels=C)
    pm = MotionModule(C)
    pm.load_state_dict(om.state_dict(), strict=True)
    pm.to(dev)
    refm = om(x)
    with ops.split_k(False):
        ops.COUNTER = {"flops": 0, "launches": 0}
        outm = pm.run(Ctx(dtype, b, F), xt)
        n_fused = ops.COUNTER["launches"]
        ops.BLOCK_HEAD_FUSED = False
        try:
            ops.COUNTER = {"flops": 0, "launches": 0}
            outm_u = pm.run(Ctx(dtype, b, F), xt)
            n_unfused = ops.COUNTER["launches"]
        finally:
            ops.BLOCK_HEAD_FUSED = True
            ops.COUNTER = None
    assert n_fused < n_unfused, (n_fused, n_unfused)
    e = rel_l2(from_tok(outm.cpu(), b, F) - x, refm - x)
    eu = rel_l2(outm - xt, outm_u - xt)
    report(f"motion module C320 16x16 F{F} heads + tail fused {dtype}: rel_l2(branch)={e:.2e} (vs unfused heads {eu:.2e}; {n_fused} vs {n_unfused} launches)")
    assert e < TOL[dtype] and eu < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_pose_guider(dev, dtype):
    og, pg = build_pair_pose(dtype, dev)
    x = torch.rand(1, 3, 3, 32, 40, generator=torch.Generator().manual_seed(4))
    ref = og(x)
    out = pg(x.to(dev)).cpu()
    e = rel_l2(out, ref)
    report(f"pose guider {dtype}: rel_l2={e:.2e}")
    assert e < 2 * TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("H,W", [(64, 64), (48, 80)])  # 48x80: 60 mid-block tokens, not a multiple of 8 (784^2 case)
def test_vae_encode_decode(dev, dtype, H, W):
    """Both precision policies of the VAE (vae.py): "split" (hi + lo operand pairs; the encoder's default) must meet the
    north star's 1e-3 in isolation, "half" (plain 16-bit operands; the decoder's default) is reported under its guard."""
    ov, pv = build_pair_vae(dtype, dev)
    assert (pv.encode_precision, pv.decode_precision) == ("split", "half")
    g = torch.Generator().manual_seed(5)
    img = torch.rand(2, 3, H, W, generator=g) * 2 - 1
    z = torch.randn(2, 4, H // 8, W // 8, generator=g)
    ref_m = ov.encode(img).latent_dist.mean
    ref_d = ov.decode(z).sample
    err = {}
    for pol in ("split", "half"):
        pv.encode_precision = pv.decode_precision = pol
        err["encode", pol] = rel_l2(pv.encode(img.to(dev)).latent_dist.mean.cpu(), ref_m)
        err["decode", pol] = rel_l2(pv.decode(z.to(dev)).sample.cpu(), ref_d)
    pv.encode_precision, pv.decode_precision = "split", "half"
    report(f"vae {dtype} {H}x{W}: split policy encode rel_l2={err['encode', 'split']:.2e} decode rel_l2={err['decode', 'split']:.2e} | "
           f"half policy encode rel_l2={err['encode', 'half']:.2e} decode rel_l2={err['decode', 'half']:.2e}")
    # split policy: ~22-bit operands whatever the 16-bit format — the bar holds for fp16 AND bf16
    assert err["encode", "split"] < 1e-3 and err["decode", "split"] < 1e-3, err
    # half policy, measured fp16 (round 5): encode 1.47e-3 / 1.55e-3, decode 2.14e-3 / 2.01e-3; bf16 scales with its mantissa.
    # The decoder's DEFAULT is "half" (full-size sd-vae-ft-mse decode measures 4.2e-4 at 784x784, tests/test_golden.py);
    # on these half-width random-weight models it misses the bar, which is reported as such.
    lim = {torch.float16: (1.85e-3, 2.5e-3), torch.bfloat16: (1.55e-2, 2.0e-2)}[dtype]  # <= 1.2 x the measured figures
    assert err["encode", "half"] < lim[0], err
    north_star(report, f"half-width VAE alone {H}x{W} {dtype}, DECODE under its default policy 'half' (not a denoised-latents figure; "
               f"policy 'split' measures {err['decode', 'split']:.2e})", {"decode": err["decode", "half"]}, {"decode": lim[1]},
               "16-bit MFMA operands: fp16 weight rounding alone is 1.4e-3 and the 3x3-conv operands 1.2e-3 on the decoder "
               "(profiles/r4_error_budget_vae.txt); decode_precision = 'split' meets the bar at 3x the decoder's MFMA work"
               if dtype == torch.float16 else "bf16 operands: 8 mantissa bits (stated limit, DESIGN.md section 4)")


def test_split3_operands_carry_the_fp32_product(dev):
    """ops.split3 + packing.pack_*_split3 through the ordinary GEMM / conv kernels against an fp64 product: ~1e-6, where plain
    fp16 operands give ~3e-4; the GroupNorm + SiLU form against torch; the zero padding of ld > 3C."""
    from mimo_amd import ops
    from mimo_amd.packing import pack_conv, pack_conv_split3, pack_linear_split3
    g = torch.Generator().manual_seed(9)
    for dtype, tol in ((torch.float16, 3e-6), (torch.bfloat16, 6e-5)):
        a = torch.randn(512, 64, generator=g).to(dev)
        w = torch.randn(96, 64, generator=g).to(dev)
        ref = a.double() @ w.double().t()
        out = ops.gemm(ops.split3(a, dtype=dtype), pack_linear_split3(w, dtype), out_f32=True)
        plain = ops.gemm(a.to(dtype), w.to(dtype).contiguous(), out_f32=True)
        e3, e1 = rel_l2(out.cpu(), ref.cpu()), rel_l2(plain.cpu(), ref.cpu())
        assert e3 < tol and e1 > 20 * e3, (dtype, e3, e1)
        # 3x3 conv with GroupNorm + SiLU in front and a fused 1x1 shortcut segment
        x = torch.randn(2, 16, 24, 32, generator=g).to(dev)               # [n, H, W, C]
        cw = (torch.randn(48, 32, 3, 3, generator=g) * 0.1).to(dev)
        sw = (torch.randn(48, 32, 1, 1, generator=g) * 0.1).to(dev)
        gam, bet = torch.randn(32, generator=g).to(dev), torch.randn(32, generator=g).to(dev)
        st = ops.group_norm_stats(x, groups=8, eps=1e-6, dtype=dtype)
        a3 = ops.split3(x, st, gam, bet, groups=8, silu=True, dtype=dtype)
        y = ops.conv2d(a3, pack_conv_split3(cw, dtype, shortcut=sw), 48, x2=ops.split3(x, dtype=dtype), out_f32=True)
        xn = x.permute(0, 3, 1, 2).double()
        rn = torch.nn.functional.silu(torch.nn.functional.group_norm(xn, 8, gam.double(), bet.double(), 1e-6))
        rc = torch.nn.functional.conv2d(rn, cw.double(), padding=1) + torch.nn.functional.conv2d(xn, sw.double())
        e = rel_l2(y.permute(0, 3, 1, 2).cpu(), rc.cpu())
        assert e < 20 * tol, (dtype, e)   # (the device's fast exp in SiLU and the fp32 statistics are part of this figure)
        pad = ops.split3(x[..., :8].contiguous(), dtype=dtype, ld=32)
        assert pad.shape[-1] == 32 and bool((pad[..., 24:] == 0).all()) and torch.equal(pad[..., :8], pad[..., 8:16])
        assert torch.equal(pad[..., :8], x[..., :8].to(dtype))


def test_frames_differ_and_encoder_dedup(dev):
    """mimo_frames_differ flags run starts; Pose2VideoPipeline._encode_frames encodes one frame per run of identical frames
    and returns exactly what the frame-by-frame encode returns."""
    from mimo_amd import ops
    from mimo_amd.pipeline import Pose2VideoPipeline
    from mimo_amd.scheduler import DDIMScheduler
    from oracle import synth
    g = torch.Generator().manual_seed(12)
    a, b = torch.rand(3, 32, 32, generator=g) * 2 - 1, torch.rand(3, 32, 32, generator=g) * 2 - 1
    c = b.clone()
    c[2, 31, 31] += 0.25  # one element, the last position of the frame
    frames = torch.stack([a, a, a, b, b, c, a]).to(dev)
    assert ops.frames_differ(frames).tolist() == [1, 0, 0, 1, 0, 1, 1]
    assert ops.frames_differ(frames[:1].contiguous()).tolist() == [1]
    assert ops.frames_differ(torch.zeros(4, 3, device=dev)) is None  # 12-byte frames: not comparable in 16-byte pieces
    _, pv = build_pair_vae(torch.float16, dev)
    pipe = Pose2VideoPipeline(pv, None, None, None, None, DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS))
    for pol in ("split", "half"):
        pv.encode_precision = pol
        pipe.vae_batch = 8
        ops.COUNTER = {"flops": 0, "launches": 0}
        lat = pipe._encode_frames(frames)
        dedup_flops = ops.COUNTER["flops"]
        pipe.dedup_frames = False
        ops.COUNTER = {"flops": 0, "launches": 0}
        with ops.split_k(False):
            full = pipe._encode_frames(frames)
            all_flops, ops.COUNTER = ops.COUNTER["flops"], None
            pipe.dedup_frames = True
            lat_bi = pipe._encode_frames(frames)
        assert lat.shape == full.shape == (7, 4, 4, 4)
        assert torch.equal(lat_bi, full), pol      # batch-invariant launches (split-K off): bit for bit
        assert rel_l2(lat.cpu(), full.cpu()) < 1e-5, pol
        assert torch.equal(lat[0], lat[1]) and torch.equal(lat[3], lat[4]) and not torch.equal(lat[4], lat[5])
        assert abs(dedup_flops / all_flops - 4 / 7) < 1e-6  # 4 runs of 7 frames


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("H,W,rows", [(64, 64, 16), (64, 64, 24), (48, 80, 10), (128, 96, 32)])
def test_vae_tiled_decode_is_bit_identical(dev, dtype, H, W, rows):
    """BASELINE configs[4] "VAE tiled decode": row-band tiling with exact halos (GroupNorm-apply + 3x3 conv per band, the
    nearest-x2 up-sampling convs per band of the up-sampled image, statistics global per image) must reproduce the
    untiled decode BIT FOR BIT — the reference has no blended tiling (pipeline :82-86).  Band heights that do not divide the
    image, odd band counts and non-square images included."""
    _, pv = build_pair_vae(dtype, dev)
    g = torch.Generator().manual_seed(7)
    z = torch.randn(2, 4, H // 8, W // 8, generator=g).to(dev)
    ref = pv.decode(z).sample
    pv.enable_tiling(rows)
    try:
        out = pv.decode(z).sample
    finally:
        pv.disable_tiling()
    assert torch.equal(out, ref), float((out.float() - ref.float()).abs().max())


def _run_oracle_unet(o3, o2, x, t, ehs, pose, ref_lat):
    from oracle import models as OM
    w = OM.ReferenceAttentionControl(o2, "write")
    r = OM.ReferenceAttentionControl(o3, "read")
    o2(ref_lat.repeat(2, 1, 1, 1), torch.zeros(()), ehs)
    r.update(w)
    out = o3(x, t, ehs, pose_cond_fea=pose)
    banks = [b.bank[0].clone() for b in o3.spatial_blocks()]
    r.clear()
    w.clear()
    return out, banks


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hw,F", [(16, 8), (13, 3)])
def test_denoising_unet_forward_with_reference_bank(dev, dtype, hw, F):
    """Whole denoising-UNet forward (read mode, CFG batch) + reference-UNet bank capture vs the oracle,
    through the reference-compatible forward() surfaces."""
    from mimo_amd.unet import ReferenceAttentionControl
    o3, o2, p3, p2 = build_pair_unets(dtype, dev)
    g = torch.Generator().manual_seed(6)
    ehs = torch.cat([torch.zeros(1, 1, 768), torch.randn(1, 1, 768, generator=g)])
    ref_lat = torch.randn(1, 4, hw, hw, generator=g)
    x = torch.randn(2, 8, F, hw, hw, generator=g)
    pose = torch.randn(2, 160, F, hw, hw, generator=g)
    t = 749
    with torch.no_grad():
        ref, banks = _run_oracle_unet(o3, o2, x, torch.tensor(t), ehs, pose, ref_lat)
    w = ReferenceAttentionControl(p2, mode="write", do_classifier_free_guidance=True)
    r = ReferenceAttentionControl(p3, mode="read", do_classifier_free_guidance=True)
    p2(ref_lat.repeat(2, 1, 1, 1).to(dev), 0, ehs.to(dev), stop_after=w.last_block())
    r.update(w)
    for ob, pb in zip(banks, p3.spatial_blocks()):  # bank parity (cond row), incl. pairing order
        assert rel_l2(pb.bank[0][0].float().cpu(), ob[1].float()) < TOL[dtype]
    out = p3(x.to(dev), t, ehs.to(dev), pose_cond_fea=pose.to(dev), return_dict=False)[0].cpu()
    e = rel_l2(out, ref)
    e_u, e_c = rel_l2(out[0], ref[0]), rel_l2(out[1], ref[1])
    report(f"denoising unet fwd hw{hw} F{F} {dtype}: rel_l2={e:.2e} (uncond {e_u:.2e}, cond {e_c:.2e})")
    assert e < {torch.float16: 3e-3, torch.bfloat16: 3e-2}[dtype]
    # the uncond half must not depend on the bank (mutual_self_attention.py:189-197)
    for blk in p3.spatial_blocks():
        blk.bank_kv = None
    out2 = p3(x.to(dev), t, ehs.to(dev), pose_cond_fea=pose.to(dev), return_dict=False)[0].cpu()
    assert torch.equal(out2[0], out[0]) and not torch.equal(out2[1], out[1])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hw,F", [(16, 8), (13, 3)])
def test_denoising_unet_forward_split_policy(dev, dtype, hw, F):
    """The same forward under `precision = "split"` (mimo_amd.precise: hi + lo operand pairs through the ordinary GEMM / conv
    kernels, unfused): what is left is the 16-bit Q / K / V, P and attention output — the north star's 1e-3 holds for fp16 AND
    bf16 (default policy: 7.4e-4 / 5.8e-3 on this case)."""
    from mimo_amd.unet import ReferenceAttentionControl
    o3, o2, p3, p2 = build_pair_unets(dtype, dev)
    p3.precision = p2.precision = "split"
    g = torch.Generator().manual_seed(6)
    ehs = torch.cat([torch.zeros(1, 1, 768), torch.randn(1, 1, 768, generator=g)])
    ref_lat = torch.randn(1, 4, hw, hw, generator=g)
    x = torch.randn(2, 8, F, hw, hw, generator=g)
    pose = torch.randn(2, 160, F, hw, hw, generator=g)
    t = 749
    with torch.no_grad():
        ref, banks = _run_oracle_unet(o3, o2, x, torch.tensor(t), ehs, pose, ref_lat)
    w = ReferenceAttentionControl(p2, mode="write", do_classifier_free_guidance=True)
    r = ReferenceAttentionControl(p3, mode="read", do_classifier_free_guidance=True)
    p2(ref_lat.repeat(2, 1, 1, 1).to(dev), 0, ehs.to(dev), stop_after=w.last_block())
    r.update(w)
    out = p3(x.to(dev), t, ehs.to(dev), pose_cond_fea=pose.to(dev), return_dict=False)[0].cpu()
    e = rel_l2(out, ref)
    report(f"denoising unet fwd hw{hw} F{F} {dtype} SPLIT policy: rel_l2={e:.2e}")
    assert e < {torch.float16: 3e-4, torch.bfloat16: 1e-3}[dtype], e
    for blk in p3.spatial_blocks():   # the uncond half must not depend on the bank here either
        blk.bank_kv = None
    out2 = p3(x.to(dev), t, ehs.to(dev), pose_cond_fea=pose.to(dev), return_dict=False)[0].cpu()
    assert torch.equal(out2[0], out[0]) and not torch.equal(out2[1], out[1])


@pytest.mark.parametrize("dtype", DTYPES)
def test_pipeline_split_policy_meets_the_bar_with_guidance(dev, dtype):
    """The guidance-3.5 clip the default policy misses the bar on (F = 26, two wrapped windows, 2 steps: latents 1.21e-3 fp16,
    9.6e-3 bf16) under the split policy for both UNets and both VAE directions: latents and decoded video within 1e-3 of the
    fp32 oracle — in fp16 and in bf16."""
    from mimo_amd.pipeline import Pose2VideoPipeline
    from mimo_amd.scheduler import DDIMScheduler
    from oracle import primitives as OP, synth
    from oracle.pipeline import run_clip
    o3, o2, p3, p2 = build_pair_unets(dtype, dev, seed=61)
    ov, pv = build_pair_vae(dtype, dev, seed=62)
    og, pg = build_pair_pose(dtype, dev, seed=63)
    p3.precision = p2.precision = "split"
    pv.decode_precision = "split"
    H = W = 64
    F = 26
    g = torch.Generator().manual_seed(7)
    ref_img = torch.rand(1, 3, H, W, generator=g) * 2 - 1
    bk = torch.rand(F, 3, H, W, generator=g) * 2 - 1
    pose = torch.rand(F, 3, H, W, generator=g)
    clip = torch.randn(1, 768, generator=g)
    lat = torch.randn(1, 4, F, H // 8, W // 8, generator=g)
    with torch.no_grad():
        vid_o, lat_o = run_clip(ov, o2, o3, og, OP.DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS), clip, ref_img, bk, pose, lat, 2, 3.5)
    pipe = Pose2VideoPipeline(pv, None, p2, p3, pg, DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS))
    vid_p, lat_p = pipe.run_tensors(ref_img.to(dev), bk.to(dev), pose.to(dev), clip.to(dev), lat.to(dev), 2, 3.5, return_latents=True)
    e_lat, e_vid = rel_l2(lat_p.cpu(), lat_o), rel_l2(vid_p.cpu(), vid_o)
    report(f"pipeline F26 2 steps guidance 3.5 {dtype} SPLIT policy (UNets + VAE): latents rel_l2={e_lat:.2e} video rel_l2={e_vid:.2e}")
    if dtype == torch.float16:
        assert e_lat < 1e-3 and e_vid < 1e-3, (e_lat, e_vid)   # measured 1.8e-4 / 1.0e-4
    else:
        # bf16: what the split policy leaves in 16 bits — Q / K / V, the probabilities and the attention output — carries 8
        # mantissa bits: 5.7e-4 on one forward (within the bar, test above), 1.45e-3 once guidance 3.5 weighs the two branches
        assert e_vid < 1e-3
        north_star(report, "half-width models, F = 26, 2 steps, guidance 3.5, bf16 under the SPLIT policy", {"latents": e_lat}, 1.75e-3,
                   "bf16 Q / K / V, P and attention outputs (the 16-bit tensors the split policy keeps) x CFG amplification; "
                   "one forward measures 5.7e-4; the default bf16 policy measures 1.13e-2 here")


@pytest.mark.parametrize("dtype", DTYPES)
def test_pipeline_two_wrapped_windows_vs_oracle(dev, dtype):
    """run_tensors (VAE encode, pose guider, reference UNet, 2 DDIM steps x 2 wrapped windows x CFG, VAE decode)
    vs oracle.pipeline.run_clip on identical injected latents: final latents and decoded video."""
    from mimo_amd.pipeline import Pose2VideoPipeline
    from mimo_amd.scheduler import DDIMScheduler
    from oracle import primitives as OP, synth
    from oracle.pipeline import run_clip
    o3, o2, p3, p2 = build_pair_unets(dtype, dev, seed=61)
    ov, pv = build_pair_vae(dtype, dev, seed=62)
    og, pg = build_pair_pose(dtype, dev, seed=63)
    H = W = 64
    F = 26
    g = torch.Generator().manual_seed(7)
    ref_img = torch.rand(1, 3, H, W, generator=g) * 2 - 1
    bk = torch.rand(F, 3, H, W, generator=g) * 2 - 1
    pose = torch.rand(F, 3, H, W, generator=g)
    clip = torch.randn(1, 768, generator=g)
    lat = torch.randn(1, 4, F, H // 8, W // 8, generator=g)
    with torch.no_grad():
        vid_o, lat_o = run_clip(ov, o2, o3, og, OP.DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS), clip, ref_img, bk, pose,
                                lat, 2, 3.5)
    pipe = Pose2VideoPipeline(pv, None, p2, p3, pg, DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS))
    vid_p, lat_p = pipe.run_tensors(ref_img.to(dev), bk.to(dev), pose.to(dev), clip.to(dev), lat.to(dev), 2, 3.5,
                                    return_latents=True)
    e_lat, e_vid = rel_l2(lat_p.cpu(), lat_o), rel_l2(vid_p.cpu(), vid_o)
    report(f"pipeline F26 2 steps {dtype}: latents rel_l2={e_lat:.2e} video rel_l2={e_vid:.2e}")
    assert vid_p.shape == (1, 3, F, H, W)
    # CFG at guidance 3.5 (see test_pipeline_edge_cases_vs_oracle for the bound and the bisect).  Regression guards = 1.2 x the
    # round-6 measurements (fp16 1.21e-3 / 8.6e-4; bf16 9.6e-3 / 6.8e-3: 8x the rounding, cannot meet 1e-3 at all)
    guard = {torch.float16: {"latents": 1.45e-3, "video": 1.03e-3}, torch.bfloat16: {"latents": 1.16e-2, "video": 8.2e-3}}[dtype]
    north_star(report, f"half-width models, F = 26, 2 steps, guidance 3.5, {dtype}", {"latents": e_lat, "video": e_vid}, guard,
               "CFG amplification u + 3.5 (c - u) on random-weight models (profiles/r4_edge_case_bisect.txt)" if dtype == torch.float16
               else "bf16 operands: 8 mantissa bits (stated limit, DESIGN.md section 4)")


@pytest.mark.parametrize("F,guidance,hw", [(1, 3.5, 64), (5, 1.0, 64), (24, 1.0, 40), (3, 3.5, 104), (5, 1.0, (48, 104)), (4, 1.0, (72, 40))])
def test_pipeline_edge_cases_vs_oracle(dev, F, guidance, hw):
    """Edges of run_tensors against oracle.pipeline.run_clip: a ONE-frame clip (temporal attention over a single frame),
    guidance 1.0 (no CFG: b = 1 forwards, banks read by every row, the no-CFG quirk of cfg_ddim), a full 24-frame window
    without CFG on a 40x40 image (5x5 latents: every level below the 32-row MFMA tile), a 104x104 image (13x13 latents:
    odd sizes 13 / 7 / 4 / 2 through every down- and explicit-size up-sampler), and two NON-SQUARE clips (height x width 48x104 =
    6x13 latents, 72x40 = 9x5: every kernel takes H and W apart; odd and even sizes mixed per axis) — fp16."""
    from mimo_amd.pipeline import Pose2VideoPipeline
    from mimo_amd.scheduler import DDIMScheduler
    from oracle import primitives as OP, synth
    from oracle.pipeline import run_clip
    dtype = torch.float16
    o3, o2, p3, p2 = build_pair_unets(dtype, dev, seed=81)
    ov, pv = build_pair_vae(dtype, dev, seed=82)
    og, pg = build_pair_pose(dtype, dev, seed=83)
    H, W = hw if isinstance(hw, tuple) else (hw, hw)
    hw = f"{H}x{W}" if H != W else hw
    g = torch.Generator().manual_seed(9)
    ref_img = torch.rand(1, 3, H, W, generator=g) * 2 - 1
    bk = torch.rand(F, 3, H, W, generator=g) * 2 - 1
    pose = torch.rand(F, 3, H, W, generator=g)
    clip = torch.randn(1, 768, generator=g)
    lat = torch.randn(1, 4, F, H // 8, W // 8, generator=g)
    with torch.no_grad():
        vid_o, lat_o = run_clip(ov, o2, o3, og, OP.DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS), clip, ref_img, bk, pose, lat, 2, guidance)
    pipe = Pose2VideoPipeline(pv, None, p2, p3, pg, DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS))
    vid_p, lat_p = pipe.run_tensors(ref_img.to(dev), bk.to(dev), pose.to(dev), clip.to(dev), lat.to(dev), 2, guidance, return_latents=True)
    e_lat, e_vid = rel_l2(lat_p.cpu(), lat_o), rel_l2(vid_p.cpu(), vid_o)
    hw = f"{hw}x{hw}" if H == W else hw
    report(f"pipeline edge F={F} guidance={guidance} {hw} fp16: latents rel_l2={e_lat:.2e} video rel_l2={e_vid:.2e}")
    assert vid_p.shape == (1, 3, F, H, W) and bool(torch.isfinite(vid_p).all())
    # Bars, not measurements: 1e-3 without CFG.  With guidance w the step combines u + w (c - u): the two branches' rounding
    # errors enter with weights (1 - w) and w while the result stays of the size of one branch (these random-weight models barely
    # react to the CLIP embedding), i.e. up to sqrt(w^2 + (w - 1)^2) = 4.3x at w = 3.5 for independent errors — 2e-3.  The
    # bisect (profiles/r4_edge_case_bisect.txt: the same runs fed the ORACLE's VAE latents and pose features) shows the
    # 1.2-1.6e-3 of the CFG cases is this, not the VAE or the pose guider: exact inputs leave it unchanged, guidance 1 gives 6.8e-4.
    # (round 4 measured: F=1 g=3.5 1.59e-3 / 1.10e-3; F=5 g=1 6.8e-4 / 6.3e-4; F=24 g=1 6.5e-4 / 6.1e-4; F=3 g=3.5 1.20e-3 / 7.4e-4)
    if guidance == 1.0:
        assert e_lat < 1.0e-3 and e_vid < 1.0e-3
    else:
        # the same clip under the split precision policy (UNets + VAE decode): within the bar
        p3.precision = p2.precision = "split"
        pv.decode_precision = "split"
        vid_s, lat_s = pipe.run_tensors(ref_img.to(dev), bk.to(dev), pose.to(dev), clip.to(dev), lat.to(dev), 2, guidance, return_latents=True)
        es_lat, es_vid = rel_l2(lat_s.cpu(), lat_o), rel_l2(vid_s.cpu(), vid_o)
        report(f"pipeline edge F={F} guidance={guidance} {hw} fp16 SPLIT policy: latents rel_l2={es_lat:.2e} video rel_l2={es_vid:.2e}")
        assert es_lat < 1e-3 and es_vid < 1e-3
        # regression guards = 1.2 x the round-6 measurements (F = 1: 1.20e-3 / 9.4e-4; F = 3: 1.12e-3 / 6.8e-4)
        guard = {1: {"latents": 1.44e-3, "video": 1.13e-3}, 3: {"latents": 1.35e-3, "video": 8.2e-4}}[F]
        north_star(report, f"half-width models edge case F = {F}, guidance {guidance}, {hw} fp16, DEFAULT policy (split policy: "
                   f"{es_lat:.2e})", {"latents": e_lat, "video": e_vid},
                   guard, "CFG amplification u + 3.5 (c - u) on random-weight models (profiles/r4_edge_case_bisect.txt); 6.8e-4 without CFG")


@pytest.mark.parametrize("F,ctx", [(20, dict(context_frames=8, context_stride=1, context_overlap=2)),
                                   (20, dict(context_frames=8, context_stride=2, context_overlap=4)),
                                   (13, dict(context_frames=16, context_stride=1, context_overlap=4))])
def test_pipeline_context_schedule_parameters_vs_oracle(dev, F, ctx):
    """Non-default window schedules (context.py:15-42 through pipeline :505-553) against oracle.pipeline.run_clip: windows of 8
    frames with overlap 2 (frames covered by one, two windows), context_stride 2 (a second pass of windows over every other frame:
    strided frame indices, frames covered up to four times), and a clip SHORTER than context_frames (one window of all 13 frames) —
    3 DDIM steps (the schedule depends on the step index), guidance 1, fp16."""
    from mimo_amd.pipeline import Pose2VideoPipeline
    from mimo_amd.scheduler import DDIMScheduler
    from mimo_amd import context as CX
    from oracle import primitives as OP, synth
    from oracle.pipeline import run_clip
    dtype = torch.float16
    o3, o2, p3, p2 = build_pair_unets(dtype, dev, seed=91)
    ov, pv = build_pair_vae(dtype, dev, seed=92)
    og, pg = build_pair_pose(dtype, dev, seed=93)
    H = W = 64
    g = torch.Generator().manual_seed(19)
    ref_img = torch.rand(1, 3, H, W, generator=g) * 2 - 1
    bk = torch.rand(F, 3, H, W, generator=g) * 2 - 1
    pose = torch.rand(F, 3, H, W, generator=g)
    clip = torch.randn(1, 768, generator=g)
    lat = torch.randn(1, 4, F, H // 8, W // 8, generator=g)
    with torch.no_grad():
        vid_o, lat_o = run_clip(ov, o2, o3, og, OP.DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS), clip, ref_img, bk, pose, lat, 3, 1.0, **ctx)
    pipe = Pose2VideoPipeline(pv, None, p2, p3, pg, DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS))
    vid_p, lat_p = pipe.run_tensors(ref_img.to(dev), bk.to(dev), pose.to(dev), clip.to(dev), lat.to(dev), 3, 1.0, return_latents=True, **ctx)
    e_lat, e_vid = rel_l2(lat_p.cpu(), lat_o), rel_l2(vid_p.cpu(), vid_o)
    nwin = [len(CX.uniform(i, 3, F, ctx["context_frames"], ctx["context_stride"], ctx["context_overlap"])) for i in range(3)]
    report(f"pipeline F={F} {ctx} ({nwin} windows per step) 3 steps guidance 1.0 fp16: latents rel_l2={e_lat:.2e} video rel_l2={e_vid:.2e}")
    assert vid_p.shape == (1, 3, F, H, W) and bool(torch.isfinite(vid_p).all())
    assert e_lat < 1.0e-3 and e_vid < 1.0e-3


def test_pipeline_rejects_what_the_reference_cannot_run(dev):
    """context windows longer than the motion modules' positional table (32) and context_batch_size > 1 fail loudly."""
    from mimo_amd.pipeline import Pose2VideoPipeline
    from mimo_amd.scheduler import DDIMScheduler
    from oracle import synth
    dtype = torch.float16
    _, _, p3, p2 = build_pair_unets(dtype, dev, seed=81)
    _, pv = build_pair_vae(dtype, dev, seed=82)
    _, pg = build_pair_pose(dtype, dev, seed=83)
    pipe = Pose2VideoPipeline(pv, None, p2, p3, pg, DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS))
    F, H = 40, 64
    g = torch.Generator().manual_seed(1)
    args = (torch.rand(1, 3, H, H, generator=g).to(dev) * 2 - 1, torch.rand(F, 3, H, H, generator=g).to(dev),
            torch.rand(F, 3, H, H, generator=g).to(dev), torch.randn(1, 768, generator=g).to(dev),
            torch.randn(1, 4, F, 8, 8, generator=g).to(dev), 1, 3.5)
    with pytest.raises(ValueError, match="temporal_position_encoding_max_len"):
        pipe.run_tensors(*args, context_frames=40)
    from PIL import Image
    im = Image.new("RGB", (64, 64))
    with pytest.raises(NotImplementedError, match="context_batch_size"):
        pipe(im, [im], [im], 64, 64, 1, 1, 3.5, context_batch_size=2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("size", ["small", "vit_l_14"])
def test_clip_image_encoder_vs_transformers(dev, dtype, size):
    """SURVEY 8(f) rank 1.  Oracle = the reference's own dependency, transformers' CLIPVisionModelWithProjection, on CPU
    fp32 with seeded random weights (norm affine parameters perturbed so they are not numerically invisible)."""
    from transformers import CLIPVisionConfig
    from transformers import CLIPVisionModelWithProjection as RefCLIP
    from mimo_amd.clip import CLIPVisionModelWithProjection
    if size == "small":
        cfg = CLIPVisionConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                               image_size=56, patch_size=14, projection_dim=96)
    else:
        if dtype == torch.bfloat16:
            pytest.skip("full ViT-L/14 tower: fp16 run only")
        cfg = CLIPVisionConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                               image_size=224, patch_size=14, projection_dim=768)
    torch.manual_seed(77)
    ref = RefCLIP(cfg).eval()
    g = torch.Generator().manual_seed(78)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "norm" in n:
                p.copy_((1.0 if n.endswith("weight") else 0.0) + 0.1 * torch.randn(p.shape, generator=g))
            elif n.endswith("bias"):
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
    prod = CLIPVisionModelWithProjection(cfg)
    missing, unexpected = prod.load_state_dict(ref.state_dict(), strict=False)
    assert not unexpected and all(k.endswith("position_ids") for k in missing), (missing, unexpected)
    prod.to(dev)
    prod.compute_dtype = dtype
    B = 2 if size == "small" else 1
    x = torch.randn(B, 3, cfg.image_size, cfg.image_size, generator=g)
    with torch.no_grad():
        r = ref(pixel_values=x)
    o = prod(x.to(dev))
    e1 = rel_l2(o.image_embeds.float().cpu(), r.image_embeds)
    e2 = rel_l2(o.last_hidden_state.float().cpu(), r.last_hidden_state)
    report(f"clip image encoder {size} {dtype}: image_embeds rel_l2={e1:.2e} last_hidden rel_l2={e2:.2e}")
    assert e1 < 3 * TOL[dtype] and e2 < 3 * TOL[dtype]


def _sharded_clip_worker(rank, world, port, q, F=26, plan="cross_step"):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        q.put((rank, _small_clip(torch.device("cuda:0"), shard=True, F=F, plan=plan).cpu().numpy()))  # by value: the worker exits
    finally:
        dist.destroy_process_group()


def _rccl_world1_worker(port, q, F, plan="cross_step"):
    """One rank, backend "nccl" (= RCCL): the sharded code path forced on (pipe.shard_force), so that every collective of
    the long-clip mode — the per-slot async all_gather_into_tensor of UnitExchange on RCCL's stream, the all_gather of the
    sharded per-frame stages — actually runs on RCCL."""
    import torch.distributed as dist
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        out = _small_clip(dev, shard=True, F=F, force=True, plan=plan)
        t = torch.ones(4, device=dev)
        dist.all_reduce(t)  # the reduction bench.py takes its max-over-ranks time with
        torch.cuda.synchronize()
        q.put(("ok", out.cpu().numpy(), float(t.sum())))
    except Exception as e:  # surface the failure in the parent instead of a queue timeout
        q.put(("error", repr(e), 0.0))
    finally:
        dist.destroy_process_group()


def _small_clip(dev, shard=False, invariant=False, window_streams=None, F=26, delay_main_cycles=0, force=False, graphs=False,
                plan="cross_step", steps=2):
    from mimo_amd.pipeline import Pose2VideoPipeline
    from mimo_amd.scheduler import DDIMScheduler
    from oracle import synth
    dtype = torch.float16
    _, _, p3, p2 = build_pair_unets(dtype, dev, seed=61)
    _, pv = build_pair_vae(dtype, dev, seed=62)
    _, pg = build_pair_pose(dtype, dev, seed=63)
    H = W = 64  # F = 26: two wrapped 24-frame windows -> 4 (window, CFG-half) units; F = 50: three windows; F = 24: one
    g = torch.Generator().manual_seed(7)
    ref_img = torch.rand(1, 3, H, W, generator=g) * 2 - 1
    bk = torch.rand(F, 3, H, W, generator=g) * 2 - 1
    pose = torch.rand(F, 3, H, W, generator=g)
    clip = torch.randn(1, 768, generator=g)
    lat = torch.randn(1, 4, F, H // 8, W // 8, generator=g)
    pipe = Pose2VideoPipeline(pv, None, p2, p3, pg, DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS))
    pipe.shard_windows, pipe.batch_invariant = shard, invariant
    pipe.shard_force = force
    pipe.shard_plan = plan
    pipe.use_graphs = graphs
    if window_streams is not None:
        pipe.window_streams = window_streams
    if delay_main_cycles:  # the main stream falls far behind the host: whatever a side stream needs from it must be ordered by events
        torch.cuda._sleep(int(delay_main_cycles))
    vid, latents = pipe.run_tensors(ref_img.to(dev), bk.to(dev), pose.to(dev), clip.to(dev), lat.to(dev), steps, 3.5, return_latents=True)
    return torch.cat([latents.flatten(), vid.flatten()])


def test_window_streams_do_not_change_the_result(dev):
    """The independent windows of a step run on two HIP streams (pipeline.window_streams): same kernels, per-stream split-K
    scratch, accumulation in canonical window order -> bit-identical to the one-stream run, twice in a row (no race)."""
    one = _small_clip(dev, window_streams=1)
    two = _small_clip(dev, window_streams=2)
    again = _small_clip(dev, window_streams=2)
    assert torch.isfinite(one).all()
    assert torch.equal(one, two) and torch.equal(two, again)
    # Round-2 advisor finding: weights are packed lazily; a side stream must never consume packed buffers whose pack
    # kernels sit on another stream.  Fresh (unpacked) models + a main stream stalled ~0.2 s behind the host: the pack
    # kernels (issued on main by Pose2VideoPipeline.prepack before the fork) are still queued when the side streams'
    # launches are issued — only the wait_stream(main) events keep the result right.
    delayed = _small_clip(dev, window_streams=2, delay_main_cycles=4e8)
    assert torch.equal(one, delayed)


def test_hipgraph_replay_of_the_forward_matches_the_eager_run(dev):
    """pipe.use_graphs (bench.py --graphs): the denoising forward of a (batch, window) shape is captured once as a hipGraph and
    replayed per step.  The captured forward consumes the same per-step rows of the clip tables (time-embedding projections,
    collapsed cross-attentions) as the eager run and launches the same kernels on the same buffers: bit-identical, replay
    after replay, one window and two."""
    for F in (24, 26):
        eager = _small_clip(dev, F=F, window_streams=1)
        graph = _small_clip(dev, F=F, graphs=True)
        again = _small_clip(dev, F=F, graphs=True)
        assert torch.isfinite(graph).all() and torch.equal(graph, again)
        assert torch.equal(graph, eager)


@pytest.mark.parametrize("world,F", [(2, 26), (2, 24), (2, 50), pytest.param(4, 26, marks=pytest.mark.skipif(
    not os.environ.get("MIMO_TEST_WORLD4"), reason="four processes time-slicing one GPU take ~4 min; set MIMO_TEST_WORLD4=1 "
    "(passed on the MI355X box of round 2: profiles/r2_sharded_world4_one_gpu.txt)"))])
@pytest.mark.parametrize("plan", ["cross_step", "step_sync"])
def test_sharded_long_clip_equals_single_gpu_bit_for_bit(dev, world, F, plan):
    """one clip whose work items and per-frame stages are dealt over `world` ranks (all on this GPU,
    collectives over gloo with host staging — RCCL refuses two ranks on one device) must reproduce the single-process
    result EXACTLY (fixed canonical summation order, no atomics, split-K off on both sides).  The plan (pipeline.plan_items)
    decides what a rank runs: F = 26 (two wrapped windows) on two ranks = one whole b = 2 window each; F = 24 (one window) =
    its cond half on rank 0 and its uncond half on rank 1, as b = 1 forwards; F = 50 (three windows) = a whole window per
    rank + the third window's two halves, one per rank (whole windows and single halves mixed on one rank, two exchange
    slots); four ranks at F = 26 = one half each.  plan = "cross_step" (round 6, the default): the same clips under the slot
    schedule of plan_cross_step — F = 50 on two ranks: slots {w0, w2} whole, then w1 as its two halves, per step."""
    import torch.multiprocessing as mp
    import os
    from mimo_amd.pipeline import plan_items
    from mimo_amd.context import get_context_scheduler
    nw = len(get_context_scheduler("uniform")(0, 2, F, 24, 1, 4))
    kinds = sorted(len(it) for r in plan_items(nw, True, world) for it in r)
    assert kinds == {(2, 26): [2, 2], (2, 24): [1, 1], (2, 50): [1, 1, 2, 2], (4, 26): [1, 1, 1, 1]}[(world, F)]
    single = _small_clip(dev, invariant=True, F=F).cpu()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 400 + 11 + world
    port += F
    port += 50 if plan == "step_sync" else 0
    procs = [ctx.Process(target=_sharded_clip_worker, args=(r, world, port, q, F, plan)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = {r: torch.from_numpy(v) for r, v in (q.get(timeout=900) for _ in procs)}
    for p_ in procs:
        p_.join(timeout=120)
    assert torch.isfinite(single).all()
    assert all(torch.equal(res[r], single) for r in range(world))
    report(f"sharded long clip ({world} ranks, F = {F}, {plan} plan" + (f", item sizes {kinds}" if plan == "step_sync" else "") +
           ", 2 steps, fp16): latents and video bit-identical to the single-process run")


@pytest.mark.parametrize("plan", ["cross_step", "step_sync"])
@pytest.mark.parametrize("F", [26, 50])
def test_rccl_branch_world1_equals_plain_run_bit_for_bit(dev, F, plan):
    """The `nccl` (RCCL) branch of the long-clip mode on the one GPU of this box: a process group of ONE rank with the
    sharded path forced on runs the unit plan (whole windows as b = 2 items on alternating HIP streams), UnitExchange's
    async all_gather_into_tensor per slot on RCCL's stream, and the sharded per-frame stages' all_gather — and reproduces the
    plain single-process run bit for bit (F = 26: two windows, two exchange slots x 2 units; F = 50: three windows)."""
    import torch.multiprocessing as mp
    single = _small_clip(dev, invariant=True, F=F).cpu()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 400 + 431 + F + (7 if plan == "step_sync" else 0)
    p_ = ctx.Process(target=_rccl_world1_worker, args=(port, q, F, plan))
    p_.start()
    status, val, ar = q.get(timeout=900)
    p_.join(timeout=120)
    assert status == "ok", val
    assert ar == 4.0
    assert torch.equal(torch.from_numpy(val), single)
    report(f"RCCL branch (backend nccl, world 1, forced sharding, {plan} plan, F = {F}): latents and video bit-identical to the plain run")


def _edit_template(n=44, H=120, W=160, seed=0):
    """A synthetic video-editing template: pose frames with a textured person blob that walks right and grows (the ROI-clip
    cutter fires), random video / background frames, an occluder mask that sweeps through."""
    import numpy as np
    rs = np.random.RandomState(seed)
    pose, vid, bk, occ = [], [], [], []
    for i in range(n):
        f = np.zeros((H, W, 3), np.uint8)
        cx, cy = 30 + 2 * i, H // 2
        hw = 10 + (0 if i < 20 else 2 * (i - 20))
        hh = 25 + (0 if i < 25 else (i - 25))
        y0, y1, x0, x1 = max(0, cy - hh), min(H, cy + hh), max(0, cx - hw), min(W, cx + hw)
        f[y0:y1, x0:x1] = rs.randint(30, 255, (y1 - y0, x1 - x0, 3))
        pose.append(f)
        vid.append(rs.randint(0, 255, (H, W, 3), dtype=np.uint8))
        bk.append(rs.randint(0, 255, (H, W, 3), dtype=np.uint8))
        o = np.zeros((H, W, 3), np.uint8)
        o[:, max(0, 3 * i - 10):3 * i + 12] = 255
        o[H // 3:H // 2, :] //= 2  # soft occluder values too
        occ.append(o)
    return pose, vid, bk, occ


def test_run_edit_end_to_end_vs_oracle_chain(dev):
    """BASELINE configs[2] as ONE path: mimo_amd.run_edit.MIMO.run (run_edit.py:153-306: reference crop + pad, frame-rate
    selection + time crop, ROI-clip segmentation, per-frame padding, Pose2VideoPipeline.__call__, per-frame compositing with
    edge mask / occluder / clip cross-fade) on a synthetic template with two clips and an occluder, against the oracle
    chain — the SAME host-side template functions (bit-exact vs the reference's own tools/util.py in test_host_cpu.py),
    the CPU fp32 oracle pipeline, and oracle/edit.py (the numpy / PIL restatement of run_edit.py:253-304).
    uint8 frames: the compositing itself is bit-exact (checked here on the oracle's video); end to end the only difference
    is the fp16 pipeline, i.e. +-1 grey level where `(image * 255).astype(np.uint8)` truncates next to an integer."""
    import numpy as np
    from PIL import Image
    from transformers import CLIPImageProcessor, CLIPVisionConfig
    from transformers import CLIPVisionModelWithProjection as RefCLIP
    from mimo_amd import edit as E
    from mimo_amd.clip import CLIPVisionModelWithProjection
    from mimo_amd.edit import MASK_MODE
    from mimo_amd.pipeline import Pose2VideoPipeline
    from mimo_amd.run_edit import MIMO, Template
    from mimo_amd.scheduler import DDIMScheduler
    from oracle import edit as OE
    from oracle import primitives as OP, synth
    from oracle.pipeline import run_clip
    dtype = torch.float16
    o3, o2, p3, p2 = build_pair_unets(dtype, dev, seed=81)
    ov, pv = build_pair_vae(dtype, dev, seed=82)
    og, pg = build_pair_pose(dtype, dev, seed=83)
    ccfg = CLIPVisionConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                            image_size=224, patch_size=32, projection_dim=768)
    torch.manual_seed(5)
    oclip = RefCLIP(ccfg).eval()
    pclip = CLIPVisionModelWithProjection(ccfg)
    pclip.load_state_dict({k: v for k, v in oclip.state_dict().items() if not k.endswith("position_ids")})
    pclip.to(dev)
    pclip.compute_dtype = dtype
    pipe = Pose2VideoPipeline(vae=pv, image_encoder=pclip, reference_unet=p2, denoising_unet=p3, pose_guider=pg,
                              scheduler=DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS))
    rs = np.random.RandomState(3)
    mask_list = [rs.rand(64, 64).astype(np.float32) for _ in MASK_MODE]
    pose, vid, bk, occ = _edit_template()
    tpl = Template(vid, pose, bk, occ, fps=30, target_fps=15, time_crop={"start_idx": 2, "end_idx": 80})
    ref_img = rs.randint(0, 256, (90, 70, 3), dtype=np.uint8)
    ref_mask = np.zeros((90, 70), np.uint8)
    ref_mask[10:80, 12:60] = 255
    H = W = 64
    m = MIMO(pipe, mask_list, width=W, height=H, steps=2, cfg=3.5, seed=42)
    res, fps = m.run(ref_img, tpl, ref_mask=ref_mask)
    la = m.last
    assert fps == 15 and len(res) == m.L == 21 and len(la["context_list"]) == 2
    F = sum(len(c) for c in la["context_list"])
    assert F == len(la["pose_list"]) > m.L  # the clips overlap by `overlay` frames
    assert all(r.dtype == np.uint8 and r.shape == (120, 160, 3) for r in res)
    # ---- oracle chain on the same prepared inputs ----
    to_t = lambda im: torch.from_numpy(np.array(im.resize((W, H), Image.LANCZOS)).astype(np.float32) / 255.0).permute(2, 0, 1)
    px = CLIPImageProcessor().preprocess(la["ref_image"].resize((224, 224)), return_tensors="pt").pixel_values
    with torch.no_grad():
        emb = oclip(pixel_values=px).image_embeds
        lat = torch.randn((1, 4, F, H // 8, W // 8), generator=torch.manual_seed(42))
        vid_o, _ = run_clip(ov, o2, o3, og, OP.DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS), emb,
                            (2 * to_t(la["ref_image"]) - 1)[None], torch.stack([2 * to_t(b) - 1 for b in la["bk_list"]]),
                            torch.stack([to_t(p) for p in la["pose_list"]]), lat, 2, 3.5)
    vid_f, bk_f, occ_f = la["frames"]
    args = (la["context_list"], la["bbox_clip_list"], la["clip_pad_list"], la["clip_padv_list"], bk_f, vid_f, occ_f, la["masks"])
    res_o = OE.composite(vid_o[0].float(), *args, 4, m.L)
    # the compositing stage alone, on the oracle's video: bit-exact
    dev_o = E.composite_clips(vid_o[0].float().contiguous().to(dev), *args, overlay=4, L=m.L).cpu().numpy()
    assert all(np.array_equal(dev_o[i], res_o[i]) for i in range(m.L))
    # end to end
    d = np.abs(np.stack(res).astype(np.int32) - np.stack(res_o).astype(np.int32))
    e_vid = rel_l2(la["video"].cpu(), vid_o[0])
    report(f"run_edit MIMO.run end to end ({len(la['context_list'])} clips, {F} generated / {m.L} output frames, occluder, fp16): "
           f"video rel_l2={e_vid:.2e}; uint8 frames mean |d|={d.mean():.3f}, max |d|={int(d.max())}, "
           f"{100 * float((d > 1).mean()):.3f} % of values off by more than 1")
    assert d.mean() < 0.5 and float((d > 1).mean()) < 5e-3 and int(d.max()) <= 8
    north_star(report, "run_edit MIMO.run end to end, half-width models (decoded video, round 6: 6.9e-4)", {"video": e_vid}, 8.4e-4, CFG_CAUSE)


def test_run_animate_end_to_end_vs_oracle_chain(dev):
    """BASELINE configs[0]'s entry as ONE path: mimo_amd.run_animate.MIMO.run (run_animate.py:153-229: reference crop + pad,
    30-fps selection of the driving frames, white backgrounds, ONE human-centred crop, per-frame padding,
    Pose2VideoPipeline.__call__, uint8 frames by truncation) against the oracle chain on the same prepared inputs (the host-side
    template functions are bit-exact vs the reference's own tools/util.py in test_host_cpu.py)."""
    import numpy as np
    from PIL import Image
    from transformers import CLIPImageProcessor, CLIPVisionConfig
    from transformers import CLIPVisionModelWithProjection as RefCLIP
    from mimo_amd.clip import CLIPVisionModelWithProjection
    from mimo_amd.pipeline import Pose2VideoPipeline
    from mimo_amd.run_animate import MIMO
    from mimo_amd.scheduler import DDIMScheduler
    from oracle import primitives as OP, synth
    from oracle.pipeline import run_clip
    dtype = torch.float16
    o3, o2, p3, p2 = build_pair_unets(dtype, dev, seed=91)
    ov, pv = build_pair_vae(dtype, dev, seed=92)
    og, pg = build_pair_pose(dtype, dev, seed=93)
    ccfg = CLIPVisionConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                            image_size=224, patch_size=32, projection_dim=768)
    torch.manual_seed(6)
    oclip = RefCLIP(ccfg).eval()
    pclip = CLIPVisionModelWithProjection(ccfg)
    pclip.load_state_dict({k: v for k, v in oclip.state_dict().items() if not k.endswith("position_ids")})
    pclip.to(dev)
    pclip.compute_dtype = dtype
    pipe = Pose2VideoPipeline(vae=pv, image_encoder=pclip, reference_unet=p2, denoising_unet=p3, pose_guider=pg,
                              scheduler=DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS))
    pose, _, _, _ = _edit_template()
    rs = np.random.RandomState(4)
    ref_img = rs.randint(0, 256, (90, 70, 3), dtype=np.uint8)
    ref_mask = np.zeros((90, 70), np.uint8)
    ref_mask[10:80, 12:60] = 255
    H = W = 64
    m = MIMO(pipe, width=W, height=H, steps=2, cfg=3.5, seed=42, max_frame_num=8)
    res, fps = m.run(ref_img, pose, fps=30, ref_mask=ref_mask)
    la = m.last
    F = len(la["pose_list"])
    assert fps == 30 and len(res) == m.L == F == 8 and all(r.dtype == np.uint8 and r.shape == (H, W, 3) for r in res)
    to_t = lambda im: torch.from_numpy(np.array(im.resize((W, H), Image.LANCZOS)).astype(np.float32) / 255.0).permute(2, 0, 1)
    px = CLIPImageProcessor().preprocess(la["ref_image"].resize((224, 224)), return_tensors="pt").pixel_values
    with torch.no_grad():
        emb = oclip(pixel_values=px).image_embeds
        lat = torch.randn((1, 4, F, H // 8, W // 8), generator=torch.manual_seed(42))
        vid_o, _ = run_clip(ov, o2, o3, og, OP.DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS), emb,
                            (2 * to_t(la["ref_image"]) - 1)[None], torch.stack([2 * to_t(b) - 1 for b in la["bk_list"]]),
                            torch.stack([to_t(p) for p in la["pose_list"]]), lat, 2, 3.5)
    res_o = (vid_o[0].float().permute(1, 2, 3, 0).numpy() * 255).astype(np.uint8)   # run_animate.py:223-225
    d = np.abs(np.stack(res).astype(np.int32) - res_o.astype(np.int32))
    e_vid = rel_l2(la["video"].cpu(), vid_o[0])
    report(f"run_animate MIMO.run end to end ({F} frames, fp16): video rel_l2={e_vid:.2e}; uint8 frames mean |d|={d.mean():.3f}, "
           f"max |d|={int(d.max())}, {100 * float((d > 1).mean()):.3f} % of values off by more than 1")
    assert d.mean() < 0.5 and float((d > 1).mean()) < 5e-3 and int(d.max()) <= 8
    # the same entry under the split precision policy (UNets + both VAE directions; the CLIP encoder stays on 16-bit operands)
    p3.precision = p2.precision = "split"
    pv.decode_precision = "split"
    m2 = MIMO(pipe, width=W, height=H, steps=2, cfg=3.5, seed=42, max_frame_num=8)
    res2, _ = m2.run(ref_img, pose, fps=30, ref_mask=ref_mask)
    e_split = rel_l2(m2.last["video"].cpu(), vid_o[0])
    d2 = np.abs(np.stack(res2).astype(np.int32) - res_o.astype(np.int32))
    report(f"run_animate MIMO.run end to end under the SPLIT policy: video rel_l2={e_split:.2e}; uint8 frames max |d|={int(d2.max())}, "
           f"{100 * float((d2 > 0).mean()):.3f} % of values differ")
    assert e_split < 1e-3
    north_star(report, "run_animate MIMO.run end to end, half-width models, DEFAULT policy (decoded video; round 6: 1.04e-3; split policy: "
               f"{e_split:.2e})", {"video": e_vid}, 1.25e-3, CFG_CAUSE)


def test_run_animate_files_in_file_out(dev, tmp_path):
    """The reference's `main` shape (run_animate.py:231-249): a reference IMAGE FILE and a template DIRECTORY holding `sdc.mp4` in, an
    `.mp4` out — mimo_amd.run_animate.MIMO.run_paths over mimo_amd.video_io (Motion-JPEG in ISO-BMFF both ways, the codec this
    image has).  The frames the pipeline sees are exactly the decoded JPEG frames (same result as MIMO.run on them), and the
    written file decodes to the generated frames within JPEG error at 30 fps."""
    import numpy as np
    from PIL import Image
    from transformers import CLIPVisionConfig
    from mimo_amd import video_io as V
    from mimo_amd.clip import CLIPVisionModelWithProjection
    from mimo_amd.pipeline import Pose2VideoPipeline
    from mimo_amd.run_animate import MIMO
    from mimo_amd.scheduler import DDIMScheduler
    from oracle import synth
    dtype = torch.float16
    _, _, p3, p2 = build_pair_unets(dtype, dev, seed=91)
    _, pv = build_pair_vae(dtype, dev, seed=92)
    _, pg = build_pair_pose(dtype, dev, seed=93)
    torch.manual_seed(6)
    pclip = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                                                           image_size=224, patch_size=32, projection_dim=768))
    pclip.to(dev)
    pclip.compute_dtype = dtype
    pipe = Pose2VideoPipeline(vae=pv, image_encoder=pclip, reference_unet=p2, denoising_unet=p3, pose_guider=pg,
                              scheduler=DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS))
    pose, _, _, _ = _edit_template()
    tpl = tmp_path / "tpl"
    tpl.mkdir()
    V.save_video(pose[:12], str(tpl / "sdc.mp4"), fps=30, codec="mjpeg!", quality=95)
    rs = np.random.RandomState(4)
    ref = np.full((90, 70, 3), 255, np.uint8)
    ref[10:80, 12:60] = rs.randint(0, 256, (70, 48, 3), dtype=np.uint8)
    Image.fromarray(ref).save(str(tmp_path / "ref.png"))
    out = MIMO(pipe, width=64, height=64, steps=2, cfg=3.5, seed=42, max_frame_num=8).run_paths(
        str(tmp_path / "ref.png"), str(tpl), str(tmp_path / "result.mp4"), codec="mjpeg!", quality=98)
    frames, fps = V.read_frames(out)
    assert abs(fps - 30.0) < 1e-9 and len(frames) == 8 and frames[0].size == (64, 64)
    decoded, native = V.read_frames(str(tpl / "sdc.mp4"))
    m = MIMO(pipe, width=64, height=64, steps=2, cfg=3.5, seed=42, max_frame_num=8)
    res, _ = m.run(Image.open(str(tmp_path / "ref.png")).convert("RGB"), decoded, fps=native)
    d = np.abs(np.stack([np.asarray(f) for f in frames]).astype(np.int32) - np.stack(res).astype(np.int32))
    report(f"run_animate files in / file out (sdc.mp4 -> result.mp4, Motion-JPEG): 8 frames at {fps:.0f} fps, result vs generated frames "
           f"mean |d|={d.mean():.2f} (JPEG q98)")
    assert d.mean() < 3.0


def test_pipeline_call_surface_pil_inputs(dev):
    """The reference's call surface (run_animate.py:208-218 / run_edit.py): PIL reference image, lists of PIL pose and
    per-frame background images (config 3: non-constant backgrounds), CPU generator, `.videos` [1,3,F,H,W] float32 on the
    host in [0,1] — with the HIP CLIP encoder as `image_encoder` — against the oracle's tensor-level clip."""
    import numpy as np
    from PIL import Image
    from transformers import CLIPVisionConfig
    from transformers import CLIPVisionModelWithProjection as RefCLIP
    from mimo_amd.clip import CLIPVisionModelWithProjection
    from mimo_amd.pipeline import Pose2VideoPipeline
    from mimo_amd.scheduler import DDIMScheduler
    from oracle import primitives as OP, synth
    from oracle.pipeline import run_clip
    dtype = torch.float16
    o3, o2, p3, p2 = build_pair_unets(dtype, dev, seed=71)
    ov, pv = build_pair_vae(dtype, dev, seed=72)
    og, pg = build_pair_pose(dtype, dev, seed=73)
    ccfg = CLIPVisionConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                            image_size=224, patch_size=32, projection_dim=768)
    torch.manual_seed(5)
    oclip = RefCLIP(ccfg).eval()
    pclip = CLIPVisionModelWithProjection(ccfg)
    pclip.load_state_dict({k: v for k, v in oclip.state_dict().items() if not k.endswith("position_ids")})
    pclip.to(dev)
    pclip.compute_dtype = dtype
    H = W = 64
    F = 8
    ref_img = Image.fromarray(np.random.RandomState(0).randint(0, 256, (H, W, 3), dtype=np.uint8))
    poses = [Image.fromarray(np.random.RandomState(100 + i).randint(0, 256, (H, W, 3), dtype=np.uint8)) for i in range(F)]
    bks = [Image.fromarray(np.random.RandomState(200 + i).randint(0, 256, (H, W, 3), dtype=np.uint8)) for i in range(F)]
    pipe = Pose2VideoPipeline(vae=pv, image_encoder=pclip, reference_unet=p2, denoising_unet=p3, pose_guider=pg,
                              scheduler=DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS))
    seen = []
    out = pipe(ref_img, poses, bks, W, H, F, 2, 3.5, generator=torch.manual_seed(42),
               callback=lambda i, t, lat: seen.append(int(t)), callback_steps=1).videos
    assert out.shape == (1, 3, F, H, W) and out.dtype == torch.float32 and out.device.type == "cpu"
    assert float(out.min()) >= 0.0 and float(out.max()) <= 1.0 and len(seen) == 2
    # oracle: same preprocessing on tensors, transformers CLIP on CPU for the embedding, same injected latents
    from transformers import CLIPImageProcessor
    to_t = lambda im: torch.from_numpy(np.array(im).astype(np.float32) / 255.0).permute(2, 0, 1)
    px = CLIPImageProcessor().preprocess(ref_img.resize((224, 224)), return_tensors="pt").pixel_values
    with torch.no_grad():
        emb = oclip(pixel_values=px).image_embeds
        lat = torch.randn((1, 4, F, H // 8, W // 8), generator=torch.manual_seed(42))
        vid_o, _ = run_clip(ov, o2, o3, og, OP.DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS), emb,
                            (2 * to_t(ref_img) - 1)[None], torch.stack([2 * to_t(b) - 1 for b in bks]),
                            torch.stack([to_t(p) for p in poses]), lat, 2, 3.5)
    e = rel_l2(out, vid_o)
    report(f"pipeline __call__ (PIL inputs, HIP CLIP, per-frame backgrounds) fp16: video rel_l2={e:.2e}")
    north_star(report, "Pose2VideoPipeline.__call__ with PIL inputs, half-width models (decoded video, round 6: 9.7e-4)", {"video": e}, 1.17e-3,
               CFG_CAUSE)


def test_pipeline_call_interpolation_and_output_types(dev):
    """The rest of `__call__`'s surface (pipeline :338-365, :566-578): interpolation_factor = 2 decodes (F - 1) * 2 + 1 frames — the
    original ones where the plain call has them, the inserted ones = the decode of the interpolated latents (checked through the
    oracle's VAE); output_type "numpy" and return_dict = False hand back what the reference does; VAE slicing toggles are no-ops
    on the result."""
    import numpy as np
    from PIL import Image
    from mimo_amd import pipeline as P
    from mimo_amd.scheduler import DDIMScheduler
    from oracle import synth
    from oracle.pipeline import decode_latents
    dtype = torch.float16
    o3, o2, p3, p2 = build_pair_unets(dtype, dev, seed=71)
    ov, pv = build_pair_vae(dtype, dev, seed=72)
    og, pg = build_pair_pose(dtype, dev, seed=73)

    class Emb:   # any module returning `.image_embeds` serves as the encoder
        dtype = torch.float32
        def __call__(self, px):
            return type("O", (), {"image_embeds": torch.ones(1, 768, device=px.device) * 0.01})()
    H = W = 64
    F = 5
    rs = np.random.RandomState(3)
    ref_img = Image.fromarray(rs.randint(0, 256, (H, W, 3), dtype=np.uint8))
    poses = [Image.fromarray(rs.randint(0, 256, (H, W, 3), dtype=np.uint8)) for _ in range(F)]
    bks = [Image.fromarray(rs.randint(0, 256, (H, W, 3), dtype=np.uint8)) for _ in range(F)]
    pipe = P.Pose2VideoPipeline(vae=pv, image_encoder=Emb(), reference_unet=p2, denoising_unet=p3, pose_guider=pg,
                                scheduler=DDIMScheduler(**synth.NOISE_SCHEDULER_KWARGS))
    traj = []
    plain = pipe(ref_img, poses, bks, W, H, F, 2, 1.0, generator=torch.manual_seed(4), output_type="tensor",
                 callback=lambda i, t, lat: traj.append(lat.detach().float().cpu().clone()), callback_steps=1).videos
    with pytest.raises(TypeError, match="set_tensor_interpolation_method"):
        pipe(ref_img, poses, bks, W, H, F, 2, 1.0, generator=torch.manual_seed(4), interpolation_factor=2)
    P.set_tensor_interpolation_method(False)
    try:
        pipe.enable_vae_slicing()
        out = pipe(ref_img, poses, bks, W, H, F, 2, 1.0, generator=torch.manual_seed(4), interpolation_factor=2, output_type="numpy",
                   return_dict=False)
        pipe.disable_vae_slicing()
    finally:
        P.tensor_interpolation = None
    assert isinstance(out, np.ndarray) and out.shape == (1, 3, 2 * (F - 1) + 1, H, W) and out.dtype == np.float32
    out = torch.from_numpy(out)
    e_orig = rel_l2(out[:, :, ::2], plain)
    lat = traj[-1]
    mid = 0.5 * (lat[:, :, :-1] + lat[:, :, 1:])
    with torch.no_grad():
        vid_mid = decode_latents(ov, mid)
    e_mid = rel_l2(out[:, :, 1::2], vid_mid)
    report(f"pipeline __call__ interpolation_factor 2 (linear), numpy output: original frames vs the plain call rel_l2={e_orig:.2e}, "
           f"inserted frames vs the oracle VAE on the interpolated latents rel_l2={e_mid:.2e}")
    assert e_orig == 0.0 and e_mid < 1e-3   # (batch-invariant decode: bit-identical frames; round 6 measured 6.3e-4 on the inserted ones)

