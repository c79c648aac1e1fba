"""Kernel-level parity: every HIP entry point (through the C-ABI) vs a plain PyTorch fp32
reference of the same op, on identical half-rounded inputs.  Tolerances are rel-L2 and written
per test: fp32-out kernels only differ by accumulation order; half-out kernels add one rounding
(fp16 eps 4.9e-4, bf16 eps 3.9e-3)."""
import ctypes
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu

DTYPES = [torch.float16, torch.bfloat16]
OUT_TOL = {torch.float16: 6e-4, torch.bfloat16: 5e-3}  # one output rounding
ACC_TOL = 2e-5  # fp32 output, same rounded inputs


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


def rnd(shape, dev, dtype, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev).to(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(128, 160, 64), (300, 320, 320), (1000, 128, 200), (77, 36, 72), (4096, 640, 1280), (257, 2560, 320)])
def test_gemm_plain_and_epilogues(dev, dtype, M, N, K):
    from mimo_amd import ops
    a = rnd((M, K), dev, dtype, 1)
    w = rnd((N, K), dev, dtype, 2, K ** -0.5)  # asymmetric, non-square
    bias = rnd((N,), dev, torch.float32, 3)
    res = rnd((M, N), dev, torch.float32, 4)
    ref = a.float() @ w.float().t()
    out = ops.gemm(a, w, out_f32=True)
    assert rel_l2(out, ref) < ACC_TOL
    out = ops.gemm(a, w, bias=bias, residual=res, out_f32=True)
    assert rel_l2(out, ref + bias + res) < ACC_TOL
    out = ops.gemm(a, w, bias=bias, silu=True)
    assert out.dtype == dtype
    assert rel_l2(out.float(), F.silu(ref + bias)) < OUT_TOL[dtype]
    # per-image bias + half residual + scale
    rpi = 7 if M % 7 == 0 else M
    ib = rnd((M // rpi, N), dev, torch.float32, 5)
    resh = res.to(dtype)
    out = ops.gemm(a, w, img_bias=ib, rows_per_img=rpi, residual=resh, out_f32=True, out_scale=0.5)
    assert rel_l2(out, 0.5 * (ref + ib.repeat_interleave(rpi, 0) + resh.float())) < ACC_TOL


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_strided_a_view(dev, dtype):
    from mimo_amd import ops
    big = rnd((200, 3 * 64), dev, dtype, 1)
    w = rnd((96, 64), dev, dtype, 2, 0.1)
    out = ops.gemm(big[:, 64:128], w, out_f32=True)
    assert rel_l2(out, big[:, 64:128].float() @ w.float().t()) < ACC_TOL


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,dim", [(256, 320), (130, 64)])
def test_gemm_geglu(dev, dtype, M, dim):
    from mimo_amd import ops
    from mimo_amd.packing import pack_geglu
    inner = 4 * dim
    x = rnd((M, dim), dev, dtype, 1)
    w = rnd((2 * inner, dim), dev, dtype, 2, dim ** -0.5)
    b = rnd((2 * inner,), dev, torch.float32, 3, 0.1)
    wp, bp = pack_geglu(w, b, dtype)
    out = ops.gemm(x, wp, bias=bp, geglu=True)
    assert out.shape == (M, inner)
    h = x.float() @ w.float().t() + b
    ref = h[:, :inner] * F.gelu(h[:, inner:])
    assert rel_l2(out.float(), ref) < OUT_TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [256 * 64, 256 * 70 + 37])
def test_gemm_stream_k320(dev, dtype, M):
    """K = 320 over many rows (the level-0 QKV / GEGLU FF1 linears) takes the A-in-registers, W-streamed kernel
    (gemm_stream.hip): plain half output with and without bias, a row-strided A view, a wider output row stride, GEGLU;
    ragged last panel; the result must also equal the tiled kernel's (same products, fp32 accumulation)."""
    from mimo_amd import ops
    from mimo_amd.packing import pack_geglu
    K = 320
    big = rnd((M, K + 64), dev, dtype, 1)
    a = big[:, 32:32 + K].contiguous()
    w = rnd((960, K), dev, dtype, 2, K ** -0.5)
    b = rnd((960,), dev, torch.float32, 3)
    ref = a.float() @ w.float().t()
    out = ops.gemm(a, w)
    assert out.dtype == dtype and rel_l2(out.float(), ref) < OUT_TOL[dtype]
    out = ops.gemm(a, w, bias=b)
    assert rel_l2(out.float(), ref + b) < OUT_TOL[dtype]
    # strided A view (lda = K + 64) and an output view with a wider row stride
    ob = torch.zeros((M, 1024), device=dev, dtype=dtype)
    ops.gemm(big[:, 64:64 + K], w, bias=b, out=ob[:, :960])
    assert rel_l2(ob[:, :960].float(), big[:, 64:64 + K].float() @ w.float().t() + b) < OUT_TOL[dtype]
    assert float(ob[:, 960:].abs().max()) == 0.0
    # the tiled kernel (fp32 output keeps it off the streaming path) rounds to the same halfs
    tiled = ops.gemm(a, w, bias=b, out_f32=True).to(dtype)
    assert float((ops.gemm(a, w, bias=b).float() - tiled.float()).abs().max()) <= 2 ** -7 * float(tiled.float().abs().max())
    # GEGLU
    inner = 1280
    wg = rnd((2 * inner, K), dev, dtype, 4, K ** -0.5)
    bg = rnd((2 * inner,), dev, torch.float32, 5, 0.1)
    wp, bp = pack_geglu(wg, bg, dtype)
    out = ops.gemm(a, wp, bias=bp, geglu=True)
    h = a.float() @ wg.float().t() + bg
    assert out.shape == (M, inner) and rel_l2(out.float(), h[:, :inner] * F.gelu(h[:, inner:])) < OUT_TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(256 * 26 + 64, 1280, 1280), (256 * 44 + 17, 960, 640), (48 * 9604 // 16, 1280, 640)])
def test_gemm8_ragged_tiles_at_the_end_of_an_allocation(dev, dtype, M, N, K):
    """The 8-phase kernel (gemm8_kernel: dense, K % 128 == 0, >= 128 tiles of 256 x 256) on ragged M (M % 128 = 64 is what
    the reference's default 784 x 784 gives) and ragged N (960 = 3.75 tiles), with A and W ending exactly where their
    allocation ends: a half-tile's second DMA must be clipped by the descriptor (zero fill), not read past the tensor.
    Rows / columns of the ragged tiles are checked against fp32 separately (they are where a stale read would show)."""
    from mimo_amd import ops
    assert ((M + 255) // 256) * ((N + 255) // 256) >= 128
    # one exact-size allocation each: the tensor's last byte is the allocation's last byte
    a = torch.empty(M * K, device=dev, dtype=dtype).view(M, K).copy_(rnd((M, K), dev, dtype, 1))
    w = torch.empty(N * K, device=dev, dtype=dtype).view(N, K).copy_(rnd((N, K), dev, dtype, 2, K ** -0.5))
    bias = rnd((N,), dev, torch.float32, 3)
    res = rnd((M, N), dev, torch.float32, 4)
    ref = a.float() @ w.float().t() + bias
    out = ops.gemm(a, w, bias=bias, residual=res, out_f32=True)
    assert rel_l2(out, ref + res) < ACC_TOL
    m0, n0 = (M // 256) * 256, (N // 256) * 256
    if m0 < M:
        assert rel_l2(out[m0:], (ref + res)[m0:]) < ACC_TOL
    if n0 < N:
        assert rel_l2(out[:, n0:], (ref + res)[:, n0:]) < ACC_TOL
    outh = ops.gemm(a, w, bias=bias)
    assert outh.dtype == dtype and rel_l2(outh.float(), ref) < OUT_TOL[dtype]
    assert bool(torch.isfinite(outh.float()).all())


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_kernel_choice_is_batch_invariant_bit_for_bit(dev, dtype):
    """The dense launcher picks gemm8_kernel by tile count (a function of M = rows of the BATCH) and the tiled kernel below
    it.  A frame's bits must not depend on the batch it is computed in (sharded == single-GPU bit for bit), so the two
    kernels must agree BIT FOR BIT: both accumulate K in the same order with the same MFMA and share one epilogue.
    Computed here as: the whole batch (gemm8) vs its first quarter alone (tiled kernel), half / fp32 + residual / GEGLU /
    ragged M."""
    from mimo_amd import ops
    from mimo_amd.packing import pack_geglu
    N, K = 1280, 1280
    M = 256 * 26 + 64  # 135 tiles of 256 x 256 -> gemm8; a quarter of the rows is 35 tiles -> tiled kernel
    Mq = 256 * 6 + 64
    a = rnd((M, K), dev, dtype, 1)
    w = rnd((N, K), dev, dtype, 2, K ** -0.5)
    bias = rnd((N,), dev, torch.float32, 3)
    res = rnd((M, N), dev, torch.float32, 4)
    full = ops.gemm(a, w, bias=bias)
    part = ops.gemm(a[:Mq].contiguous(), w, bias=bias)
    assert torch.equal(full[:Mq], part)
    full = ops.gemm(a, w, bias=bias, residual=res, out_f32=True)
    part = ops.gemm(a[:Mq].contiguous(), w, bias=bias, residual=res[:Mq].contiguous(), out_f32=True)
    assert torch.equal(full[:Mq], part)
    inner = 2560
    wg = rnd((2 * inner, K), dev, dtype, 5, K ** -0.5)
    bg = rnd((2 * inner,), dev, torch.float32, 6, 0.1)
    wp, bp = pack_geglu(wg, bg, dtype)
    full = ops.gemm(a, wp, bias=bp, geglu=True)
    part = ops.gemm(a[:256 * 3].contiguous(), wp, bias=bp, geglu=True)  # 3 x 20 = 60 tiles -> tiled GEGLU kernel
    assert torch.equal(full[:256 * 3], part)


CONV_CASES = [
    # n, H, W, Cin, Cout, ks, stride, pad(t,l), out_hw, upsample_to
    (2, 16, 16, 64, 160, 3, 1, None, None, None),
    (3, 9, 7, 320, 320, 3, 1, None, None, None),       # odd spatial
    (2, 16, 16, 96, 64, 3, 2, None, None, None),        # stride 2, Cin not multiple of 64
    (2, 8, 8, 64, 128, 3, 1, None, None, (16, 16)),     # nearest x2 then conv
    (2, 13, 13, 64, 64, 3, 1, None, None, (25, 25)),    # explicit-size nearest (784 path: 13 -> 25)
    (1, 16, 16, 128, 128, 3, 2, (0, 0), (8, 8), None),  # VAE asymmetric pad (0,1,0,1) + stride 2
    (2, 8, 8, 8, 320, 3, 1, None, None, None),          # thin Cin (conv_in)
    (2, 8, 8, 320, 4, 3, 1, None, None, None),          # thin Cout (conv_out)
    (2, 8, 8, 64, 64, 1, 1, None, None, None),          # 1x1
]


def torch_conv_ref(x, w, b, ks, stride, pad, out_hw, upsample_to):
    xt = x.float().permute(0, 3, 1, 2)
    if upsample_to is not None:
        xt = F.interpolate(xt, size=upsample_to, mode="nearest")
    if pad == (0, 0) and stride == 2:
        xt = F.pad(xt, (0, 1, 0, 1))
        y = F.conv2d(xt, w.float(), b, stride=2, padding=0)
    else:
        y = F.conv2d(xt, w.float(), b, stride=stride, padding=ks // 2)
    return y.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(dev, dtype, case):
    from mimo_amd import ops
    from mimo_amd.packing import pack_conv
    n, H, W, cin, cout, ks, stride, pad, out_hw, up = case
    x = rnd((n, H, W, cin), dev, dtype, 1)
    w = rnd((cout, cin, ks, ks), dev, dtype, 2, (cin * ks * ks) ** -0.5)
    b = rnd((cout,), dev, torch.float32, 3)
    ref = torch_conv_ref(x, w, b, ks, stride, pad, out_hw, up)
    out = ops.conv2d(x, pack_conv(w, dtype), cout, ksize=ks, stride=stride, pad=pad, out_hw=out_hw,
                     upsample_to=up, bias=b, out_f32=True)
    assert out.shape == ref.shape
    assert rel_l2(out, ref) < ACC_TOL


THIN_CASES = [
    # n, H, W, Cin, Cout, stride, pad  — the direct thin-input kernel (csrc/thinconv.hip) behind mimo_conv2d
    (2, 32, 48, 8, 16, 1, None),       # pose guider conv_in (3 -> 16, Cin padded to 8)
    (2, 19, 37, 16, 16, 1, None),      # ragged: strips end inside the row, odd height
    (3, 32, 32, 16, 32, 2, None),      # stride 2
    (1, 21, 21, 32, 32, 1, None),
    (2, 16, 20, 32, 96, 2, None),      # 6 of the 8 channel tiles used
    (1, 24, 40, 8, 128, 1, None),      # VAE encoder conv_in
    (2, 16, 16, 16, 32, 2, (0, 0)),    # diffusers' asymmetric (0, 1, 0, 1) padding
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", THIN_CASES)
def test_conv2d_thin_input_direct(dev, dtype, case):
    from mimo_amd import ops
    from mimo_amd.packing import pack_conv
    n, H, W, cin, cout, stride, pad = case
    x = rnd((n, H, W, cin), dev, dtype, 1)
    w = rnd((cout, cin, 3, 3), dev, dtype, 2, (cin * 9) ** -0.5)
    b = rnd((cout,), dev, torch.float32, 3)
    out_hw = (H // 2, W // 2) if pad == (0, 0) else None
    ref = torch_conv_ref(x, w, b, 3, stride, pad, out_hw, None)
    out = ops.conv2d(x, pack_conv(w, dtype), cout, stride=stride, pad=pad, out_hw=out_hw, bias=b, out_f32=True)
    assert out.shape == ref.shape and out.dtype == torch.float32
    assert rel_l2(out, ref) < ACC_TOL
    half = ops.conv2d(x, pack_conv(w, dtype), cout, stride=stride, pad=pad, out_hw=out_hw, bias=b, silu=True)
    assert half.dtype == dtype and rel_l2(half.float(), F.silu(ref)) < OUT_TOL[dtype]
    # one image alone gives the bits it has inside the batch
    one = ops.conv2d(x[-1:].contiguous(), pack_conv(w, dtype), cout, stride=stride, pad=pad, out_hw=out_hw, bias=b, out_f32=True)
    assert torch.equal(one, out[-1:])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 16, 16, 320, 4), (1, 23, 17, 128, 4), (3, 8, 40, 64, 8)])
def test_conv3x3_thin_out_gemm_plus_tapsum(dev, dtype, shape):
    """Thin-output convolution as GEMM (input read once, weight regrouped per tap) + mimo_conv3x3_tapsum vs torch and vs the
    implicit-GEMM kernel."""
    from mimo_amd import ops
    from mimo_amd.packing import pack_conv, pack_conv_taps
    n, H, W, cin, cout = shape
    x = rnd((n, H, W, cin), dev, dtype, 1)
    w = rnd((cout - 1, cin, 3, 3), dev, dtype, 2, (cin * 9) ** -0.5)   # the last channel is padding
    b = torch.cat([rnd((cout - 1,), dev, torch.float32, 3), torch.zeros(1, device=dev)])
    ref = torch_conv_ref(x, w, b[:-1], 3, 1, None, None, None)
    out = ops.conv3x3_thin_out(x, pack_conv_taps(w, dtype, cout_pad=cout), cout, bias=b)
    assert out.shape == (n, H, W, cout) and out.dtype == torch.float32
    assert rel_l2(out[..., :-1], ref) < ACC_TOL and float(out[..., -1].abs().max()) == 0.0
    old = ops.conv2d(x, pack_conv(w, dtype, cout_pad=cout), cout, bias=b, out_f32=True)
    assert rel_l2(out, old) < ACC_TOL
    one = ops.conv3x3_thin_out(x[-1:].contiguous(), pack_conv_taps(w, dtype, cout_pad=cout), cout, bias=b)
    assert torch.equal(one, out[-1:])


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv2d_resblock_epilogue_and_fused_shortcut(dev, dtype):
    """conv2 of a ResBlock: 3x3 over h + fused 1x1 shortcut over x + per-image temb-style bias + fp32 residual."""
    from mimo_amd import ops
    from mimo_amd.packing import pack_conv
    n, H, W, cin, cout, cx = 3, 8, 8, 128, 160, 192
    h = rnd((n, H, W, cin), dev, dtype, 1)
    x = rnd((n, H, W, cx), dev, dtype, 2)
    w = rnd((cout, cin, 3, 3), dev, dtype, 3, (9 * cin) ** -0.5)
    ws = rnd((cout, cx, 1, 1), dev, dtype, 4, cx ** -0.5)
    b = rnd((cout,), dev, torch.float32, 5)
    temb = rnd((n, cout), dev, torch.float32, 6)
    res = rnd((n, H, W, cout), dev, torch.float32, 7)
    ref = torch_conv_ref(h, w, b, 3, 1, None, None, None) + \
        F.conv2d(x.float().permute(0, 3, 1, 2), ws.float()).permute(0, 2, 3, 1) + temb[:, None, None, :] + res
    out = ops.conv2d(h, pack_conv(w, dtype, shortcut=ws), cout, x2=x, bias=b, img_bias=temb, residual=res,
                     out_f32=True)
    assert rel_l2(out, ref) < ACC_TOL
    # img_bias as a column slice of a wider matrix, one row per group of 3 images (batch element)
    wide = rnd((1, 3 * cout), dev, torch.float32, 8)
    out2 = ops.conv2d(h, pack_conv(w, dtype, shortcut=ws), cout, x2=x, bias=b, img_bias=wide[:, cout:2 * cout],
                      imgs_per_bias_row=3, residual=res, out_f32=True)
    assert rel_l2(out2, ref - temb[:, None, None, :] + wide[:, None, None, cout:2 * cout]) < ACC_TOL
    out = ops.conv2d(h, pack_conv(w, dtype), cout, bias=b, silu=True)
    assert rel_l2(out.float(), F.silu(torch_conv_ref(h, w, b, 3, 1, None, None, None))) < OUT_TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [33, 64, 200, 257])
def test_gemm_stream_k320_few_rows_without_split_k(dev, dtype, M):
    """Round-3 advisor finding: with split-K off (sharded / batch-invariant runs) K = 320 GEMMs of ANY row count — also below
    one 256-row panel — take the streaming kernel; the sharded-vs-single tests compare it with itself.  Here: against torch."""
    from mimo_amd import ops
    from mimo_amd.packing import pack_geglu
    K = 320
    a = rnd((M, K), dev, dtype, 1)
    w = rnd((960, K), dev, dtype, 2, K ** -0.5)
    b = rnd((960,), dev, torch.float32, 3)
    ref = a.float() @ w.float().t() + b
    with ops.split_k(False):
        out = ops.gemm(a, w, bias=b)
        assert out.dtype == dtype and rel_l2(out.float(), ref) < OUT_TOL[dtype]
        inner = 1280
        wg = rnd((2 * inner, K), dev, dtype, 4, K ** -0.5)
        bg = rnd((2 * inner,), dev, torch.float32, 5, 0.1)
        wp, bp = pack_geglu(wg, bg, dtype)
        outg = ops.gemm(a, wp, bias=bp, geglu=True)
    h = a.float() @ wg.float().t() + bg
    assert outg.shape == (M, inner) and rel_l2(outg.float(), h[:, :inner] * F.gelu(h[:, inner:])) < OUT_TOL[dtype]
    # the same rows inside a larger launch give the same bits (the kernel choice does not depend on M in this mode)
    big = torch.cat([a, rnd((512, K), dev, dtype, 6)])
    with ops.split_k(False):
        assert torch.equal(ops.gemm(big, w, bias=b)[:M], out)


HCONV_CASES = [
    # n, H, W, C1, C2, Cout, gn, upsample2x
    (2, 16, 16, 64, 0, 128, True, False),      # one tile per image: every border is padding; BN = 128 configuration
    (3, 32, 48, 192, 0, 320, True, False),     # 3 chunks (odd count: the weight ring parity runs through), BN = 320
    (2, 48, 32, 64, 32, 256, True, False),     # virtual concat, 1.5 chunks (the last sub-chunk is empty), BN = 256
    (1, 64, 64, 320, 640, 320, True, False),   # an up-block conv1 of the 64 x 64 level (960 channels = the LDS table limit)
    (2, 16, 32, 128, 0, 640, True, False),     # two channel tiles of 320
    (2, 16, 16, 64, 0, 128, False, True),      # nearest x2 then conv, plain cast (Upsample3D)
    (1, 32, 32, 640, 0, 640, False, True),
]


def hconv_ref(x1, x2, ab, w, dtype, ups):
    x = x1 if x2 is None else torch.cat([x1, x2], dim=-1)
    y = x
    if ab is not None:
        y = F.silu(x * ab[:, 0][:, None, None, :] + ab[:, 1][:, None, None, :])
    yt = y.to(dtype).float().permute(0, 3, 1, 2)
    if ups:
        yt = F.interpolate(yt, scale_factor=2, mode="nearest")
    return F.conv2d(yt, w.float(), padding=1).permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", HCONV_CASES)
def test_conv3x3_fused_groupnorm_silu(dev, dtype, case):
    """mimo_conv3x3_fused (halo-tiled conv that applies GroupNorm-affine + SiLU to its own fp32 input) vs torch: the reference
    rounds the normalised activation to half exactly where the kernel does, so only the hardware exp / rcp of the SiLU
    (a few half roundings flip) and the accumulation order differ."""
    from mimo_amd import ops
    from mimo_amd.packing import pack_conv
    n, H, W, C1, C2, cout, gn, ups = case
    C = C1 + C2
    Hs, Ws = (H // 2, W // 2) if ups else (H, W)
    x1 = rnd((n, Hs, Ws, C1), dev, torch.float32, 1, 2.0) + 0.3
    x2 = rnd((n, Hs, Ws, C2), dev, torch.float32, 2, 0.7) if C2 else None
    w = rnd((cout, C, 3, 3), dev, dtype, 3, (9 * C) ** -0.5)
    ws = rnd((cout, 96, 1, 1), dev, dtype, 9, 0.1)  # a fused-shortcut segment behind the 9 taps must be ignored
    b = rnd((cout,), dev, torch.float32, 4)
    temb = rnd((n, cout), dev, torch.float32, 5)
    res = rnd((n, H, W, cout), dev, torch.float32, 6)
    ab = None
    if gn:
        gamma, beta = rnd((C,), dev, torch.float32, 7) * 0.5 + 1.0, rnd((C,), dev, torch.float32, 8, 0.3)
        stats = ops.group_norm_stats(x1, groups=32, eps=1e-5, x2=x2, dtype=dtype)
        ab = ops.group_norm_affine(stats, gamma, beta, C, groups=32)
        xc = x1 if x2 is None else torch.cat([x1, x2], dim=-1)
        gn_ref = F.group_norm(xc.permute(0, 3, 1, 2), 32, gamma, beta, 1e-5).permute(0, 2, 3, 1)
        aff = xc * ab[:, 0][:, None, None, :] + ab[:, 1][:, None, None, :]
        assert rel_l2(aff, gn_ref) < 1e-5
    ref0 = hconv_ref(x1, x2, ab, w, dtype, ups)
    tol = 2e-4
    out = ops.conv3x3_fused(x1, pack_conv(w, dtype), cout, x2=x2, ab=ab, upsample2x=ups)
    assert out.shape == ref0.shape and out.dtype == torch.float32
    assert rel_l2(out, ref0) < tol
    wp = pack_conv(w, dtype, shortcut=ws)
    want_raw = gn and not ups
    r = ops.conv3x3_fused(x1, wp, cout, x2=x2, ab=ab, bias=b, img_bias=temb, residual=res, out_scale=0.5, upsample2x=ups,
                          want_raw=want_raw)
    out2, raw = r if want_raw else (r, None)
    assert rel_l2(out2, 0.5 * (ref0 + b + temb[:, None, None, :] + res)) < tol
    if want_raw:
        xc = x1 if x2 is None else torch.cat([x1, x2], dim=-1)
        assert torch.equal(raw, xc.to(dtype))
    # a per-batch-element bias row shared by groups of images; one image alone gives the bits it has inside the batch
    if n % 2 == 0 or n == 1:
        g = 2 if n % 2 == 0 else 1
        out3 = ops.conv3x3_fused(x1, wp, cout, x2=x2, ab=ab, img_bias=temb[: n // g], imgs_per_bias_row=g, upsample2x=ups)
        assert rel_l2(out3, ref0 + temb[: n // g].repeat_interleave(g, 0)[:, None, None, :]) < tol
    if gn and cout % 32 == 0:
        # tile statistics of the output (per 16 x 16 tile and channel, from the epilogue) merge to the GroupNorm statistics a
        # pass over the tensor computes; the output itself is the same bits with and without them
        o_ts = ops.conv3x3_fused(x1, wp, cout, x2=x2, ab=ab, bias=b, img_bias=temb, out_scale=0.5, tile_stats=True)
        o_pl = ops.conv3x3_fused(x1, wp, cout, x2=x2, ab=ab, bias=b, img_bias=temb, out_scale=0.5)
        assert ops.stats_of(o_ts) is not None and ops.stats_of(o_ts).shape == (n * (H // 16) * (W // 16), 2, cout)
        assert torch.equal(o_ts, o_pl)
        st_merge = ops.group_norm_stats(o_ts, groups=32, eps=1e-5, dtype=dtype)
        st_pass = ops.group_norm_stats(o_pl, groups=32, eps=1e-5, dtype=dtype)
        ref_mean = o_pl.view(n, H * W, 32, cout // 32).double().mean(dim=(1, 3))
        ref_var = o_pl.view(n, H * W, 32, cout // 32).double().var(dim=(1, 3), unbiased=False)
        assert float((st_merge[..., 0].double() - ref_mean).abs().max()) < 1e-5 * (1 + float(ref_mean.abs().max()))
        assert rel_l2(st_merge[..., 1], (ref_var + 1e-5).rsqrt()) < 1e-5 and rel_l2(st_pass[..., 1], st_merge[..., 1]) < 1e-5
    if cout % 32 == 0 and (gn or ups):
        def check_stats(t, plain, rows):
            cs = ops.stats_of(t)
            assert cs is not None and cs.shape == (n * H * W // rows, 2, cout) and torch.equal(t, plain)
            st = ops.group_norm_stats(t, groups=32, eps=1e-5, dtype=dtype)
            v = plain.view(n, H * W, 32, cout // 32).double()
            ref_mean, ref_var = v.mean(dim=(1, 3)), v.var(dim=(1, 3), unbiased=False)
            assert float((st[..., 0].double() - ref_mean).abs().max()) < 1e-5 * (1 + float(ref_mean.abs().max()))
            assert rel_l2(st[..., 1], (ref_var + 1e-5).rsqrt()) < 1e-5
            return cs
        if gn:
            # with a residual (the ResBlock's second convolution): statistics of the STORED values per 32-pixel slab
            kw = dict(x2=x2, ab=ab, bias=b, residual=res, out_scale=0.5)
            cs_a = check_stats(ops.conv3x3_fused(x1, wp, cout, tile_stats=True, **kw), ops.conv3x3_fused(x1, wp, cout, **kw), 32)
        else:
            # the up-sampling convolution (no affine): tile statistics from the accumulators
            kw = dict(bias=b, upsample2x=True)
            cs_a = check_stats(ops.conv3x3_fused(x1, wp, cout, tile_stats=True, **kw), ops.conv3x3_fused(x1, wp, cout, **kw), 256)
        # a virtual concat whose two sources carry statistics of DIFFERENT slab sizes merges without a pass over either
        ta = ops.conv3x3_fused(x1, wp, cout, tile_stats=True, **kw)
        tb = rnd((n, H, W, 64), dev, torch.float32, 21, 1.5) - 0.2
        v = tb.view(n, H * W // 32, 32, 64).double()
        cs_b = torch.stack([v.mean(dim=2), ((v - v.mean(dim=2, keepdim=True)) ** 2).sum(dim=2)], dim=2).float().reshape(-1, 2, 64).contiguous()
        if (cout + 64) % 32 == 0:
            for first, second in ((ta, ops.with_stats(tb, cs_b)), (ops.with_stats(tb, cs_b), ta)):
                st = ops.group_norm_stats(first, groups=32, eps=1e-5, x2=second, dtype=dtype)
                cat = torch.cat([first, second], dim=-1)
                Cc = cat.shape[-1]
                vv = cat.view(n, H * W, 32, Cc // 32).double()
                assert float((st[..., 0].double() - vv.mean(dim=(1, 3))).abs().max()) < 2e-5
                assert rel_l2(st[..., 1], (vv.var(dim=(1, 3), unbiased=False) + 1e-5).rsqrt()) < 1e-5
    one = ops.conv3x3_fused(x1[-1:].contiguous(), pack_conv(w, dtype), cout, x2=None if x2 is None else x2[-1:].contiguous(),
                            ab=None if ab is None else ab[-1:].contiguous(), upsample2x=ups)
    assert torch.equal(one, out[-1:])


def test_conv3x3_fused_rejects_what_it_does_not_cover(dev):
    from mimo_amd import lib as L, ops
    from mimo_amd.packing import pack_conv
    w = pack_conv(rnd((128, 64, 3, 3), dev, torch.float16, 1), torch.float16)
    ok = torch.zeros((1, 16, 16, 64), device=dev)
    assert ops.hconv_supported(torch.zeros((1, 64, 64, 64), device=dev), 128)
    assert not ops.hconv_supported(torch.zeros((1, 98, 98, 64), device=dev), 128)      # not a multiple of the 16 x 16 tile
    assert not ops.hconv_supported(torch.zeros((1, 64, 64, 160), device=dev), 160)     # channel counts of the half-width test models
    assert not ops.hconv_supported(torch.zeros((1, 32, 32, 640), device=dev), 640)     # below HCONV_MIN_HW: the row-tiled kernel quantises better
    assert not ops.hconv_supported(torch.zeros((1, 64, 64, 1920), device=dev), 320)    # affine table beyond the LDS budget
    with pytest.raises(L.MimoHipError):
        ops.conv3x3_fused(torch.zeros((1, 24, 16, 64), device=dev), w, 128)
    with pytest.raises(L.MimoHipError):
        ops.conv3x3_fused(ok, pack_conv(rnd((96, 64, 3, 3), dev, torch.float16, 2), torch.float16), 96)
    ops.conv3x3_fused(ok, w, 128)


@pytest.mark.parametrize("dtype", DTYPES)
def test_split_k_long_reduction_few_tiles(dev, dtype):
    """Few output tiles + long K (the 8x8-level ResBlock convs, M = 48*64, K = 9*1280+640) take the split-K route:
    fp32 partial tiles in the caller's workspace, reduced in fixed order with the full epilogue.  Checked against
    the unsplit kernels (MIMO_GEMM_SPLITK=0) bit-for-bit-close and against the fp32 reference."""
    import os
    from mimo_amd import lib as L, ops
    from mimo_amd.packing import pack_conv
    n, H, W, cin, cout, cx = 48, 8, 8, 1280, 1280, 640
    h = rnd((n, H, W, cin), dev, dtype, 1)
    x = rnd((n, H, W, cx), dev, dtype, 2)
    w = rnd((cout, cin, 3, 3), dev, dtype, 3, (9 * cin) ** -0.5)
    ws = rnd((cout, cx, 1, 1), dev, dtype, 4, cx ** -0.5)
    b = rnd((cout,), dev, torch.float32, 5)
    temb = rnd((2, cout), dev, torch.float32, 6)
    res = rnd((n, H, W, cout), dev, torch.float32, 7)
    wp = pack_conv(w, dtype, shortcut=ws)
    ref = torch_conv_ref(h, w, b, 3, 1, None, None, None) + \
        F.conv2d(x.float().permute(0, 3, 1, 2), ws.float()).permute(0, 2, 3, 1) + \
        temb.repeat_interleave(24, 0)[:, None, None, :] + res

    def run():
        o1 = ops.conv2d(h, wp, cout, x2=x, bias=b, img_bias=temb, imgs_per_bias_row=24, residual=res, out_f32=True)
        o2 = ops.conv2d(h, wp, cout, x2=x, bias=b, silu=True, out_scale=0.5)
        a = h.view(-1, cin)[:, :]
        wl = rnd((1280, cin), dev, dtype, 9, cin ** -0.5)
        a4 = torch.cat([a, a, a, a], dim=1)                      # K = 5120
        w4 = torch.cat([wl, wl, wl, wl], dim=1).contiguous()
        o3 = ops.gemm(a4, w4, bias=b, residual=res.view(-1, cout).to(dtype))
        return o1, o2, o3, a4, w4

    with ops.split_k(False):  # MIMO_EPI_NO_SPLITK on every call inside: the unsplit reference
        u1, u2, u3, a4, w4 = run()
    assert ops.split_k_enabled()
    s1, s2, s3, _, _ = run()
    assert rel_l2(s1, ref) < ACC_TOL and rel_l2(s1, u1) < 1e-5
    assert rel_l2(s2.float(), u2.float()) < OUT_TOL[dtype]
    assert rel_l2(s2.float(), 0.5 * F.silu(ref - temb.repeat_interleave(24, 0)[:, None, None, :] - res)) < OUT_TOL[dtype]
    ref3 = a4.float() @ w4.float().t() + b + res.view(-1, cout).to(dtype).float()
    assert rel_l2(s3.float(), ref3) < OUT_TOL[dtype] and rel_l2(s3.float(), u3.float()) < OUT_TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("f32in", [True, False])
@pytest.mark.parametrize("C1,C2,eps,silu", [(320, 0, 1e-5, True), (1280, 640, 1e-5, True), (320, 640, 1e-6, False), (128, 0, 1e-6, True)])
def test_group_norm_concat(dev, dtype, f32in, C1, C2, eps, silu):
    from mimo_amd import ops
    n, H, W = 3, 6, 5
    idt = torch.float32 if f32in else dtype
    x1 = (rnd((n, H, W, C1), dev, torch.float32, 1) * 2 + 0.7).to(idt)
    x2 = (rnd((n, H, W, C2), dev, torch.float32, 2) - 0.3).to(idt) if C2 else None
    C = C1 + C2
    gamma = rnd((C,), dev, torch.float32, 3) * 0.1 + 1
    beta = rnd((C,), dev, torch.float32, 4) * 0.1
    out, raw = ops.group_norm(x1, gamma, beta, eps=eps, silu=silu, x2=x2, dtype=dtype, want_raw=True)
    cat = torch.cat([x1, x2], dim=-1) if C2 else x1
    ref = F.group_norm(cat.float().permute(0, 3, 1, 2), 32, gamma, beta, eps).permute(0, 2, 3, 1)
    if silu:
        ref = F.silu(ref)
    assert rel_l2(out.float(), ref) < OUT_TOL[dtype]
    assert torch.equal(raw, cat.to(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_group_norm_pixel_sliced_stats(dev, dtype):
    """Few (image, group) pairs over many pixels (VAE shapes): the statistics grid is sliced along the pixels and
    reduced in fixed order; a large common offset checks the shifted-sum variance."""
    from mimo_amd import ops
    n, H, W, C = 2, 96, 80, 128
    x = rnd((n, H, W, C), dev, torch.float32, 1) * 0.5 + 7.0
    gamma = rnd((C,), dev, torch.float32, 3) * 0.1 + 1
    beta = rnd((C,), dev, torch.float32, 4) * 0.1
    out, _ = ops.group_norm(x, gamma, beta, eps=1e-6, silu=True, dtype=dtype)
    ref = F.silu(F.group_norm(x.permute(0, 3, 1, 2), 32, gamma, beta, 1e-6).permute(0, 2, 3, 1))
    assert rel_l2(out.float(), ref) < OUT_TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C", [320, 640, 1280])
def test_layer_norm_and_pe(dev, dtype, C):
    from mimo_amd import ops
    Fr, HW = 5, 6
    x = rnd((2 * Fr * HW, C), dev, torch.float32, 1) * 1.5 + 0.2
    g = rnd((C,), dev, torch.float32, 2) * 0.1 + 1
    b = rnd((C,), dev, torch.float32, 3) * 0.1
    out = ops.layer_norm(x, g, b, dtype=dtype)
    ref = F.layer_norm(x, (C,), g, b, 1e-5)
    assert rel_l2(out.float(), ref) < OUT_TOL[dtype]
    pe = rnd((32, C), dev, torch.float32, 4)
    out = ops.layer_norm(x, g, b, dtype=dtype, pe=pe, rows_per_frame=HW, pe_frames=Fr)
    fidx = (torch.arange(2 * Fr * HW, device=dev) // HW) % Fr
    assert rel_l2(out.float(), ref + pe[fidx]) < OUT_TOL[dtype]



@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hw,C1,C2", [(32, 320, 0), (32, 640, 320), (16, 1280, 640), (16, 320, 320)])
def test_group_norm_from_epilogue_column_stats(dev, dtype, hw, C1, C2):
    """GroupNorm statistics merged from the column statistics that the PRODUCING conv / GEMM epilogue emitted
    (mimo_epilogue_ext.colstats -> mimo_group_norm_stats_cols) equal a pass over the tensor.  The virtual concat puts
    group boundaries inside a tensor's column range (C1 = 1280, C2 = 640: 60 channels per group)."""
    from mimo_amd import ops
    from mimo_amd.packing import pack_conv
    n = 9 if hw >= 32 else 3
    a = rnd((n, hw, hw, 64), dev, dtype, 1)
    res = rnd((n, hw, hw, C1), dev, torch.float32, 5) * 2 + 0.7
    w1 = pack_conv(rnd((C1, 64, 3, 3), dev, torch.float32, 2) * 0.05, dtype)
    b1 = rnd((C1,), dev, torch.float32, 3)
    x1 = ops.conv2d(a, w1, C1, bias=b1, residual=res, out_f32=True, colstats=True)
    assert ops.stats_of(x1) is not None and ops.stats_of(x1).shape == (n * hw * hw // 32, 2, C1)
    x2 = None
    if C2:  # the second tensor comes from a dense GEMM (the proj_out / motion-module producers)
        wl = rnd((C2, 64), dev, dtype, 6, 0.2)
        x2f = ops.gemm(a.view(-1, 64), wl, bias=rnd((C2,), dev, torch.float32, 7), out_f32=True, colstats=hw * hw)
        assert ops.stats_of(x2f) is not None
        x2 = ops.with_stats(x2f.view(n, hw, hw, C2), ops.stats_of(x2f))
    C = C1 + C2
    gamma = rnd((C,), dev, torch.float32, 8) * 0.1 + 1
    beta = rnd((C,), dev, torch.float32, 9) * 0.1
    fused, _ = ops.group_norm(x1, gamma, beta, eps=1e-5, silu=True, x2=x2, dtype=dtype)
    plain, _ = ops.group_norm(x1.clone(), gamma, beta, eps=1e-5, silu=True, x2=None if x2 is None else x2.clone(), dtype=dtype)
    cat = torch.cat([x1, x2], dim=-1) if C2 else x1
    ref = F.silu(F.group_norm(cat.float().permute(0, 3, 1, 2), 32, gamma, beta, 1e-5).permute(0, 2, 3, 1))
    assert rel_l2(fused.float(), ref) < OUT_TOL[dtype]
    assert rel_l2(fused.float(), plain.float()) < OUT_TOL[dtype]
    # the statistics epilogue stores exactly what the ordinary epilogue stores
    x1b = ops.conv2d(a, w1, C1, bias=b1, residual=res, out_f32=True)
    assert torch.equal(x1, x1b)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,K,with_res,with_pe", [(4096, 320, True, True), (1000, 320, True, False), (8192, 1280, False, True), (130, 64, True, False)])
def test_gemm_fused_layer_norm_output(dev, dtype, M, K, with_res, with_pe):
    """LayerNorm (+ positional table) of the GEMM result, produced in the same epilogue as a second (half) output at
    N = 320: the nn.LayerNorm that follows proj_in / to_out / the motion-module attention outputs."""
    from mimo_amd import ops
    N, HW, Fr = 320, 128, 4
    a = rnd((M, K), dev, dtype, 1)
    w = rnd((N, K), dev, dtype, 2, K ** -0.5)
    b = rnd((N,), dev, torch.float32, 3)
    res = rnd((M, N), dev, torch.float32, 4) * 1.5 + 0.2 if with_res else None
    g = rnd((N,), dev, torch.float32, 5) * 0.1 + 1
    be = rnd((N,), dev, torch.float32, 6) * 0.1
    pe = rnd((32, N), dev, torch.float32, 7) if with_pe else None
    ln = dict(gamma=g, beta=be, eps=1e-5)
    if with_pe:
        ln.update(pe=pe, rows_per_frame=HW, pe_frames=Fr)
    out, y = ops.gemm(a, w, bias=b, residual=res, out_f32=True, ln=ln)
    ref_out = a.float() @ w.float().t() + b + (res if with_res else 0)
    assert rel_l2(out, ref_out) < ACC_TOL
    ref_y = F.layer_norm(ref_out, (N,), g, be, 1e-5)
    if with_pe:
        ref_y = ref_y + pe[(torch.arange(M, device=dev) // HW) % Fr]
    assert y.dtype == dtype and rel_l2(y.float(), ref_y) < OUT_TOL[dtype]
    # same fp32 output as the call without the fused LayerNorm
    assert rel_l2(out, ops.gemm(a, w, bias=b, residual=res, out_f32=True)) < 1e-6
    # other widths fall back to mimo_layer_norm behind the same interface
    w2 = rnd((640, K), dev, dtype, 8, K ** -0.5)
    g2, be2 = rnd((640,), dev, torch.float32, 9) * 0.1 + 1, rnd((640,), dev, torch.float32, 10) * 0.1
    o2, y2 = ops.gemm(a, w2, out_f32=True, ln=dict(gamma=g2, beta=be2))
    assert rel_l2(y2.float(), F.layer_norm(a.float() @ w2.float().t(), (640,), g2, be2, 1e-5)) < OUT_TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,K,with_res,with_pe,with_ib", [(4096, 640, True, True, False), (1000, 640, True, False, True), (8192 + 64, 2560, False, True, False),
                                                          (70, 64, True, False, False), (49152, 640, True, False, True)])
def test_gemm_fused_layer_norm_output_n640(dev, dtype, M, K, with_res, with_pe, with_ib):
    """The same at N = 640 (level 1): gemm_ln640_kernel — 64 x 640 whole-row tiles, K-tile 32, 3-deep ring — fp32 output (+ bias,
    per-image bias, residual) and LayerNorm (+ positional table) as the second, half output; ragged M; equal to the unfused pair."""
    from mimo_amd import ops
    N, HW, Fr = 640, 128, 4
    a = rnd((M, K), dev, dtype, 1)
    w = rnd((N, K), dev, dtype, 2, K ** -0.5)
    b = rnd((N,), dev, torch.float32, 3)
    res = rnd((M, N), dev, torch.float32, 4) * 1.5 + 0.2 if with_res else None
    g = rnd((N,), dev, torch.float32, 5) * 0.1 + 1
    be = rnd((N,), dev, torch.float32, 6) * 0.1
    pe = rnd((32, N), dev, torch.float32, 7) if with_pe else None
    rpi = 250 if M == 1000 else 1024
    ib = rnd(((M + rpi - 1) // rpi, N), dev, torch.float32, 8) if with_ib else None
    ln = dict(gamma=g, beta=be, eps=1e-5)
    if with_pe:
        ln.update(pe=pe, rows_per_frame=HW, pe_frames=Fr)
    kw = dict(bias=b, residual=res, out_f32=True, img_bias=ib, rows_per_img=rpi if with_ib else 0)
    ops.LN_OUT_640 = True   # (off by default: correct but slower than the two launches, see mimo_amd/ops.py)
    try:
        out, y = ops.gemm(a, w, ln=ln, **kw)
    finally:
        ops.LN_OUT_640 = False
    ref_out = a.float() @ w.float().t() + b + (res if with_res else 0)
    if with_ib:
        ref_out = ref_out + ib[torch.arange(M, device=dev) // rpi]
    assert rel_l2(out, ref_out) < ACC_TOL
    ref_y = F.layer_norm(ref_out, (N,), g, be, 1e-5)
    if with_pe:
        ref_y = ref_y + pe[(torch.arange(M, device=dev) // HW) % Fr]
    assert y.dtype == dtype and rel_l2(y.float(), ref_y) < OUT_TOL[dtype]
    for sl in (slice(0, 1), slice(M - 1, M), slice(M // 2, M // 2 + 1)):
        assert rel_l2(y[sl].float(), ref_y[sl]) < 2 * OUT_TOL[dtype]
    for c in (0, 79, 80, 319, 320, 639):
        assert rel_l2(y[:, c].float(), ref_y[:, c]) < 2 * OUT_TOL[dtype] and rel_l2(out[:, c], ref_out[:, c]) < 10 * ACC_TOL
    # the unfused pair behind the same interface (the default)
    out_u, y_u = ops.gemm(a, w, ln=ln, **kw)
    assert rel_l2(out, out_u) < 1e-6 and rel_l2(y.float(), y_u.float()) < OUT_TOL[dtype]
    # a row's bits do not depend on the launch it is computed in
    half = (M // 2) // 64 * 64
    if half and not with_pe and not with_ib:
        ops.LN_OUT_640 = True
        try:
            o2, y2 = ops.gemm(a[:half].contiguous(), w, ln=ln, bias=b, residual=None if res is None else res[:half].contiguous(), out_f32=True)
        finally:
            ops.LN_OUT_640 = False
        assert torch.equal(o2, out[:half]) and torch.equal(y2, y[:half])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,C,Nout,geglu,with_res,with_ib,with_pe", [
    (8192, 1280, 3840, False, True, False, False),      # producer and consumer on the 8-phase kernel (64-column slots)
    (8192 + 77, 640, 1920, False, True, True, False),   # 80-column slots, ragged M, per-image bias in the producer
    (4096, 640, 5120, True, False, False, False),       # GEGLU consumer
    (1024, 1280, 3840, False, True, False, True),       # few rows: tiled producer (NR = 4) and consumer; positional table
    (8192, 640, 1920, False, True, False, True),        # positional table through the 8-phase consumer (paired stores)
    (200, 1280, 10240, True, True, False, False),
])
def test_gemm_layer_norm_folded_into_consumer(dev, dtype, M, C, Nout, geglu, with_res, with_ib, with_pe):
    """C = 640 / 1280 (no tile holds a row): the LayerNorm between two GEMMs is folded into the second one — the producer leaves
    half rows + per-row partial sums (mimo_epilogue_ext row_half / row_stats), the consumer multiplies the raw rows by gamma o W
    and applies mean / rstd in its epilogue (a_row_stats / a_colsum).  Against torch fp32, against the unfolded launches, and
    bit-identical rows whatever launch (row count, hence tile height / kernel) computes them."""
    from mimo_amd import ops
    from mimo_amd.packing import pack_geglu, pack_ln_fold
    K, HW, Fr = 640, 128, 4
    TOL = 1.8 * OUT_TOL[dtype]   # three independent roundings against the fp32 reference: operand, weight, output
    a = rnd((M, K), dev, dtype, 1)
    w = rnd((C, K), dev, dtype, 2, K ** -0.5)
    b = rnd((C,), dev, torch.float32, 3)
    res = rnd((M, C), dev, torch.float32, 4) * 1.5 + 0.3 if with_res else None
    g = rnd((C,), dev, torch.float32, 5) * 0.1 + 1
    be = rnd((C,), dev, torch.float32, 6) * 0.1
    rpi = 1000
    ib = rnd(((M + rpi - 1) // rpi, C), dev, torch.float32, 8) if with_ib else None
    w2 = rnd((Nout, C), dev, torch.float32, 9, C ** -0.5)
    b2 = rnd((Nout,), dev, torch.float32, 10) * 0.2
    pe = rnd((Fr, C), dev, torch.float32, 11) * 0.5 if with_pe else None
    kw = dict(bias=b, residual=res, out_f32=True, img_bias=ib, rows_per_img=rpi if with_ib else 0)
    assert ops.ln_foldable(C)
    out, u = ops.gemm(a, w, ln=dict(gamma=g, beta=be, eps=1e-5, fold=True), **kw)
    assert isinstance(u, ops.LnFold) and u.xh.dtype == dtype and u.stats.shape == (M, ops.row_stat_slots(C), 2)
    ref_out = a.float() @ w.float().t() + b + (res if with_res else 0)
    if with_ib:
        ref_out = ref_out + ib[torch.arange(M, device=dev) // rpi]
    assert rel_l2(out, ref_out) < ACC_TOL
    # the side outputs: the stored rows rounded once, and their sums
    assert torch.equal(u.xh, out.to(dtype))
    xs = u.xh.double()
    assert rel_l2(u.stats[..., 0].double().sum(1), xs.sum(1)) < 1e-5 and rel_l2(u.stats[..., 1].double().sum(1), (xs * xs).sum(1)) < 1e-5
    # the plain launch stores the same fp32 rows
    ops.LN_FOLD = False
    try:
        out_u, y_u = ops.gemm(a, w, ln=dict(gamma=g, beta=be, eps=1e-5), **kw)
    finally:
        ops.LN_FOLD = True
    assert torch.equal(out, out_u)
    # consumer
    if geglu:
        w2p32, b2p = pack_geglu(w2, b2, torch.float32)
        f = pack_ln_fold(w2p32, g, be, b2p, dtype)
        w2u, b2u = pack_geglu(w2, b2, dtype)
    else:
        f = pack_ln_fold(w2, g, be, b2, dtype)
        w2u, b2u = w2.to(dtype).contiguous(), b2
    ckw = {}
    if with_pe:  # the table is added BEHIND the LayerNorm: through the projection it is a per-image bias row
        pew = (pe.double() @ w2.double().t()).float()
        ckw = dict(img_bias=pew[(torch.arange(M // HW, device=dev)) % Fr].contiguous(), rows_per_img=HW)
    z = ops.gemm(u, f["w"], bias=f["bias"], colsum=f["colsum"], geglu=geglu, **ckw)
    n = F.layer_norm(ref_out, (C,), g, be, 1e-5)
    if with_pe:
        n = n + pe[(torch.arange(M, device=dev) // HW) % Fr]
    h = n @ w2.t() + b2
    ref = h[:, :Nout // 2] * F.gelu(h[:, Nout // 2:]) if geglu else h
    assert z.dtype == dtype and rel_l2(z.float(), ref) < TOL
    for sl in (slice(0, 1), slice(M - 1, M), slice(M // 2, M // 2 + 1)):
        assert rel_l2(z[sl].float(), ref[sl]) < 2 * TOL
    # the unfolded launches: LayerNorm output (+ table) rounded, then the projection
    yu = y_u if not with_pe else (y_u.float() + pe[(torch.arange(M, device=dev) // HW) % Fr]).to(dtype)
    z_u = ops.gemm(yu, w2u, bias=b2u, geglu=geglu)
    assert rel_l2(z_u.float(), ref) < TOL and rel_l2(z.float(), z_u.float()) < 2 * TOL
    # a row's bits do not depend on the launch it is computed in (fewer rows: another tile height / kernel of the same family)
    half = 256 if M > 256 else 64
    o2, u2 = ops.gemm(a[:half].contiguous(), w, ln=dict(gamma=g, beta=be, eps=1e-5, fold=True), bias=b,
                      residual=None if res is None else res[:half].contiguous(), out_f32=True,
                      img_bias=ib, rows_per_img=rpi if with_ib else 0)
    assert torch.equal(o2, out[:half]) and torch.equal(u2.xh, u.xh[:half]) and torch.equal(u2.stats, u.stats[:half])
    z2 = ops.gemm(u2, f["w"], bias=f["bias"], colsum=f["colsum"], geglu=geglu,
                  **({k: (v[:half // HW].contiguous() if k == "img_bias" else v) for k, v in ckw.items()}))
    assert torch.equal(z2, z[:half])


def test_gemm_folded_layer_norm_rejects_what_it_does_not_cover(dev):
    from mimo_amd import lib as L, ops
    dtype = torch.float16
    a, w = rnd((256, 640), dev, dtype, 1), rnd((640, 640), dev, dtype, 2)
    assert ops.row_stat_slots(640) == 8 and ops.row_stat_slots(1280) == 20 and ops.row_stat_slots(100) == 0
    # half output, GEGLU / SiLU producers and convolutions have no row-statistics epilogue
    out = torch.empty((256, 640), device=dev, dtype=dtype)
    xh, st = torch.empty((256, 640), device=dev, dtype=dtype), torch.empty((256, 8, 2), device=dev)
    ext = ops._ext(row_half=xh, row_stats=st)
    ws = ops._workspace(dev)
    with pytest.raises(L.MimoHipError):
        L.call("mimo_gemm_ext", 0, a.data_ptr(), 640, w.data_ptr(), out.data_ptr(), 640, 256, 640, 640, None, None, 0, 0, None, 0,
               1.0, 0, ws.data_ptr(), ws.numel() * 4, ctypes.byref(ext), None)
    # a consumer needs both the statistics and the column sums, and an even slot count
    o32 = torch.empty((256, 640), device=dev)
    bad = L.EpilogueExt()
    bad.a_row_stats, bad.a_slots = st.data_ptr(), 8
    with pytest.raises(L.MimoHipError):
        L.call("mimo_gemm_ext", 0, a.data_ptr(), 640, w.data_ptr(), o32.data_ptr(), 640, 256, 640, 640, None, None, 0, 0, None, 0,
               1.0, L.EPI_OUT_F32, ws.data_ptr(), ws.numel() * 4, ctypes.byref(bad), None)


def sdpa_ref(q, k, v, heads):
    B, Nq, C = q.shape
    d = C // heads
    qh = q.float().reshape(B, Nq, heads, d).transpose(1, 2)
    kh = k.float().reshape(B, -1, heads, d).transpose(1, 2)
    vh = v.float().reshape(B, -1, heads, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(qh, kh, vh)
    return o.transpose(1, 2).reshape(B, Nq, C)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("d,N,Nb", [(40, 256, 256), (80, 100, 100), (160, 64, 64), (40, 130, 70), (64, 200, 0)])
def test_attention_self_plus_bank(dev, dtype, d, N, Nb):
    """cond rows (b >= 2) attend [self || bank]; uncond rows (b < 2) attend self — one launch."""
    from mimo_amd import ops
    heads, B = 4, 4
    C = heads * d
    qkv = rnd((B, N, 3 * C), dev, dtype, 1)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    if Nb:
        bank = rnd((Nb, 2 * C), dev, dtype, 2)
        k2, v2 = bank[:, :C], bank[:, C:]
        out = ops.attention(q, k, v, heads, k2=k2, v2=v2, seg2_first_batch=2)
        ref_u = sdpa_ref(q[:2], k[:2], v[:2], heads)
        kc = torch.cat([k[2:], k2[None].expand(2, -1, -1)], dim=1)
        vc = torch.cat([v[2:], v2[None].expand(2, -1, -1)], dim=1)
        ref = torch.cat([ref_u, sdpa_ref(q[2:], kc, vc, heads)], dim=0)
    else:
        out = ops.attention(q, k, v, heads)
        ref = sdpa_ref(q, k, v, heads)
    assert rel_l2(out.float(), ref) < 2.5 * OUT_TOL[dtype]  # P is rounded to half before P.V


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [128, 8192 + 77, 33, 40000])
def test_ff_fused_c320(dev, dtype, M):
    """mimo_ff_fused (C = 320): residual + GEGLU(a W1^T + b1) W2^T + b2 in one launch vs (i) a plain torch fp32 reference on
    the same half-rounded operands with the hidden activations rounded to half where the kernel rounds them, and (ii) the
    two-launch path it replaces (mimo_gemm GEGLU + mimo_gemm residual).  Ragged M (not a multiple of the 128-row panel), a
    single partial panel, more panels than CUs."""
    from mimo_amd import ops
    from mimo_amd.packing import pack_ff2_kperm, pack_geglu
    C = 320
    a = rnd((M, C), dev, dtype, 1)
    w1 = rnd((8 * C, C), dev, torch.float32, 2, C ** -0.5)
    b1 = rnd((8 * C,), dev, torch.float32, 3, 0.1)
    w2 = rnd((C, 4 * C), dev, torch.float32, 4, (4 * C) ** -0.5)
    b2 = rnd((C,), dev, torch.float32, 5, 0.1)
    res = rnd((M, C), dev, torch.float32, 6)
    w1p, b1p = pack_geglu(w1, b1, dtype)
    w2k = pack_ff2_kperm(w2, dtype)
    out = ops.ff_fused(a, w1p, b1p, w2k, b2, res)
    assert out.shape == (M, C) and out.dtype == dtype
    # reference on the rounded operands
    w1r, w2r = w1.to(dtype).float(), w2.to(dtype).float()
    hcat = a.float() @ w1r.t() + b1
    hid = (hcat[:, :4 * C] * F.gelu(hcat[:, 4 * C:])).to(dtype).float()   # the kernel rounds the hidden chunk to half
    ref = res + hid @ w2r.t() + b2
    assert rel_l2(out.float(), ref) < OUT_TOL[dtype]
    two = ops.gemm(ops.gemm(a, w1p, bias=b1p, geglu=True), w2.to(dtype).contiguous(), bias=b2, residual=res)
    assert rel_l2(out.float(), two.float()) < OUT_TOL[dtype]
    # asymmetric check of the row / column mapping: one row, one hidden unit
    assert rel_l2(out[M // 2].float(), ref[M // 2]) < 2 * OUT_TOL[dtype] and rel_l2(out[:, 7].float(), ref[:, 7]) < 2 * OUT_TOL[dtype]


def _check_tail_colstats(ops, run, plain, M, hw, dtype):
    """A fused tail called with colstats=<rows per image> writes the same output bits and attaches per-32-row-slab column
    statistics that merge to the GroupNorm statistics a pass over the output computes."""
    out = run(colstats=hw)
    cs = ops.stats_of(out)
    C = plain.shape[1]
    assert cs is not None and cs.shape == (M // 32, 2, C) and torch.equal(out, plain)
    v = plain.view(M // 32, 32, C).double()
    assert float((cs[:, 0].double() - v.mean(dim=1)).abs().max()) < 1e-5 * (1 + float(plain.abs().max()))
    assert rel_l2(cs[:, 1], ((v - v.mean(dim=1, keepdim=True)) ** 2).sum(dim=1)) < 1e-5
    img = ops.with_stats(out.view(M // hw, hw, 1, C), cs)
    st = ops.group_norm_stats(img, groups=32, eps=1e-6, dtype=dtype)
    vv = plain.view(M // hw, hw, 32, C // 32).double()
    assert float((st[..., 0].double() - vv.mean(dim=(1, 3))).abs().max()) < 1e-5 * (1 + float(plain.abs().max()))
    assert rel_l2(st[..., 1], (vv.var(dim=(1, 3), unbiased=False) + 1e-6).rsqrt()) < 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [128, 8192 + 77, 33, 40000, 4096])
def test_ff_proj_fused_c320(dev, dtype, M):
    """mimo_ff_proj_fused (C = 320): x + (residual + FF(a)) Wp^T + bp in one launch vs the three launches it replaces and a
    torch fp32 reference with the kernel's two half roundings (hidden activations, FF result)."""
    from mimo_amd import ops
    from mimo_amd.packing import pack_ff2_kperm, pack_geglu, pack_proj_tail
    C = 320
    a = rnd((M, C), dev, dtype, 1)
    w1 = rnd((8 * C, C), dev, torch.float32, 2, C ** -0.5)
    b1 = rnd((8 * C,), dev, torch.float32, 3, 0.1)
    w2 = rnd((C, 4 * C), dev, torch.float32, 4, (4 * C) ** -0.5)
    b2 = rnd((C,), dev, torch.float32, 5, 0.1)
    res = rnd((M, C), dev, torch.float32, 6)
    wp = rnd((C, C), dev, torch.float32, 7, C ** -0.5)
    bp = rnd((C,), dev, torch.float32, 8, 0.1)
    x = rnd((M, C), dev, torch.float32, 9)
    w1p, b1p = pack_geglu(w1, b1, dtype)
    out = ops.ff_proj_fused(a, w1p, b1p, pack_ff2_kperm(w2, dtype), b2, res, pack_proj_tail(wp, dtype), bp, x)
    assert out.shape == (M, C) and out.dtype == torch.float32
    hcat = a.float() @ w1.to(dtype).float().t() + b1
    hid = (hcat[:, :4 * C] * F.gelu(hcat[:, 4 * C:])).to(dtype).float()
    z = (res + hid @ w2.to(dtype).float().t() + b2).to(dtype).float()
    ref = x + z @ wp.to(dtype).float().t() + bp
    assert rel_l2(out, ref) < ACC_TOL * 5  # fp32 output; the half roundings are replicated in the reference
    zz = ops.gemm(ops.gemm(a, w1p, bias=b1p, geglu=True), w2.to(dtype).contiguous(), bias=b2, residual=res)
    three = ops.gemm(zz, wp.to(dtype).contiguous(), bias=bp, residual=x, out_f32=True)
    assert rel_l2(out, three) < OUT_TOL[dtype]
    assert rel_l2(out[M // 2], ref[M // 2]) < 1e-4 and rel_l2(out[:, 161], ref[:, 161]) < 1e-4 and rel_l2(out[:, 7], ref[:, 7]) < 1e-4
    if M % 128 == 0:
        _check_tail_colstats(ops, lambda **kw: ops.ff_proj_fused(a, w1p, b1p, pack_ff2_kperm(w2, dtype), b2, res, pack_proj_tail(wp, dtype), bp, x, **kw),
                             out, M, 64, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,rows_per_img", [(128, 0), (8192 + 77, 1000), (33, 0), (40000, 4096 + 64), (1024, 128)])
def test_block_tail_fused_c320(dev, dtype, M, rows_per_img):
    """mimo_block_tail_fused (C = 320): attention output projection (+ per-image vector) + residual, LayerNorm, feed-forward,
    proj_out + residual in one launch vs the four launches it replaces and a torch fp32 reference that rounds to half where
    the kernel does (LayerNorm output, hidden activations, feed-forward result).  rows_per_img that do not divide the
    128-row panel: a panel straddles two images."""
    from mimo_amd import ops
    from mimo_amd.packing import pack_block_tail_stream, pack_ff2_kperm, pack_geglu
    C = 320
    o = rnd((M, C), dev, dtype, 1)
    wo = rnd((C, C), dev, torch.float32, 10, C ** -0.5)
    bo = rnd((C,), dev, torch.float32, 11, 0.1)
    t = rnd((M, C), dev, torch.float32, 12) + 0.5   # a row mean away from zero
    gamma = 1 + rnd((C,), dev, torch.float32, 13, 0.2)
    beta = rnd((C,), dev, torch.float32, 14, 0.2)
    w1 = rnd((8 * C, C), dev, torch.float32, 2, C ** -0.5)
    b1 = rnd((8 * C,), dev, torch.float32, 3, 0.1)
    w2 = rnd((C, 4 * C), dev, torch.float32, 4, (4 * C) ** -0.5)
    b2 = rnd((C,), dev, torch.float32, 5, 0.1)
    wp = rnd((C, C), dev, torch.float32, 7, C ** -0.5)
    bp = rnd((C,), dev, torch.float32, 8, 0.1)
    x = rnd((M, C), dev, torch.float32, 9)
    ib = ibv = None
    if rows_per_img:
        nimg = (M + rows_per_img - 1) // rows_per_img
        ib = rnd((nimg, 3 * C), dev, torch.float32, 15)[:, C:2 * C]   # a row-strided view, as the pipeline hands it over
        ibv = ib[torch.arange(M, device=dev) // rows_per_img]
    w1p, b1p = pack_geglu(w1, b1, dtype)
    ws = pack_block_tail_stream(wo, w1p, wp, dtype)
    out = ops.block_tail_fused(o, ws, bo, t, gamma, beta, 1e-5, b1p, pack_ff2_kperm(w2, dtype), b2, bp, x,
                               img_bias=ib, rows_per_img=rows_per_img or 1)
    assert out.shape == (M, C) and out.dtype == torch.float32
    y = t + o.float() @ wo.to(dtype).float().t() + bo + (ibv if ibv is not None else 0)
    n = F.layer_norm(y, (C,), gamma, beta, 1e-5).to(dtype).float()
    hcat = n @ w1.to(dtype).float().t() + b1
    hid = (hcat[:, :4 * C] * F.gelu(hcat[:, 4 * C:])).to(dtype).float()
    z = (y + hid @ w2.to(dtype).float().t() + b2).to(dtype).float()
    ref = x + z @ wp.to(dtype).float().t() + bp
    # (a half rounding of n that lands on the other side of a tie moves a few elements: looser than the pure-fp32 bound)
    assert rel_l2(out, ref) < OUT_TOL[dtype] / 2
    yy, nn = ops.gemm(o, wo.to(dtype).contiguous(), bias=bo, img_bias=ib, rows_per_img=rows_per_img or 1, residual=t, out_f32=True,
                      ln=dict(gamma=gamma, beta=beta, eps=1e-5))
    zz = ops.gemm(ops.gemm(nn, w1p, bias=b1p, geglu=True), w2.to(dtype).contiguous(), bias=b2, residual=yy)
    four = ops.gemm(zz, wp.to(dtype).contiguous(), bias=bp, residual=x, out_f32=True)
    assert rel_l2(out, four) < OUT_TOL[dtype]
    assert rel_l2(out[M // 2], ref[M // 2]) < OUT_TOL[dtype] and rel_l2(out[:, 161], ref[:, 161]) < OUT_TOL[dtype]
    assert rel_l2(out[:, 7], ref[:, 7]) < OUT_TOL[dtype] and rel_l2(out[-1], ref[-1]) < OUT_TOL[dtype]
    if M % 128 == 0:
        _check_tail_colstats(ops, lambda **kw: ops.block_tail_fused(o, ws, bo, t, gamma, beta, 1e-5, b1p, pack_ff2_kperm(w2, dtype), b2, bp, x,
                                                                    img_bias=ib, rows_per_img=rows_per_img or 1, **kw),
                             out, M, 128 if rows_per_img else 64, dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,variant", [(1024, "gn"), (4096 * 3, "gn_pe"), (128, "a"), (8192 + 77, "a_res"), (33, "a_res"), (2048, "a_res_pe"),
                                       (5 * 456, "gn_pe_straddle"), (7 * 200 + 19, "a_res_pe_straddle")])
def test_block_head_fused_c320(dev, dtype, M, variant):
    """mimo_block_head_fused (C = 320): (GroupNorm-apply +) projection (+ residual) -> y (fp32, written), LayerNorm (+ positional
    table) -> QKV in one launch, vs a torch fp32 reference that rounds to half where the kernel does (the normalised input, the
    LayerNorm output) and vs the three launches it replaces (GroupNorm-apply, GEMM + fused LayerNorm, QKV GEMM).
    Variants: gn = fp32 block input + GroupNorm affine (the spatial transformer's head), gn_pe = + positional table (the motion
    module's first head), a* = half operand (an attention output) with residual / table (its second head); ragged M;
    *_straddle: rows per image / frame that do not divide the 128-row panel (panels hold rows of two images / frames)."""
    from mimo_amd import ops
    from mimo_amd.packing import pack_block_head_stream
    C = 320
    gn, res_on, pe_on = variant.startswith("gn"), "res" in variant, "pe" in variant
    wi = rnd((C, C), dev, torch.float32, 10, C ** -0.5)
    bi = rnd((C,), dev, torch.float32, 11, 0.1)
    wqkv = rnd((3 * C, C), dev, torch.float32, 12, C ** -0.5)
    gamma = 1 + rnd((C,), dev, torch.float32, 13, 0.2)
    beta = rnd((C,), dev, torch.float32, 14, 0.2)
    ws = pack_block_head_stream(wi, wqkv, dtype)
    straddle = "straddle" in variant            # image / frame sizes that do not divide the 128-row panel (784 x 784 gives 9604)
    rpi = (456 if straddle else 512) if gn else 0   # rows per image (GroupNorm affine) = rows per frame (positional table)
    rpf, frames = (rpi or (200 if straddle else 256)), 3
    pe = rnd((5, C), dev, torch.float32, 15, 0.5) if pe_on else None
    res = (rnd((M, C), dev, torch.float32, 16) + 0.5) if res_on else None
    kw = dict(residual=res, pe=pe, rows_per_frame=rpf if pe_on else 0, pe_frames=frames if pe_on else 0)
    if gn:
        x = rnd((M, C), dev, torch.float32, 1, 2.0) + 0.3
        ab = torch.stack([1 + rnd((M // rpi, C), dev, torch.float32, 2, 0.3), rnd((M // rpi, C), dev, torch.float32, 3, 0.3)], 1).contiguous()
        y, qkv = ops.block_head_fused(ws, bi, gamma, beta, 1e-5, x=x, gn_ab=ab, rows_per_img=rpi, **kw)
        img = torch.arange(M, device=dev) // rpi
        a_ref = torch.addcmul(ab[img, 1], x, ab[img, 0]).to(dtype)   # fma, one rounding to half
    else:
        a_ref = rnd((M, C), dev, dtype, 1)
        y, qkv = ops.block_head_fused(ws, bi, gamma, beta, 1e-5, a=a_ref, **kw)
    assert y.shape == (M, C) and y.dtype == torch.float32 and qkv.shape == (M, 3 * C) and qkv.dtype == dtype
    y_ref = a_ref.float() @ wi.to(dtype).float().t() + bi + (res if res_on else 0)
    # (the GroupNorm variant's operand: torch.addcmul is not guaranteed to fuse; a half tie moved either way shows up at 1e-4)
    assert rel_l2(y, y_ref) < (ACC_TOL if not gn else 3e-4)
    n = F.layer_norm(y, (C,), gamma, beta, 1e-5)
    if pe_on:
        n = n + pe[(torch.arange(M, device=dev) // rpf) % frames]
    q_ref = n.to(dtype).float() @ wqkv.to(dtype).float().t()
    assert rel_l2(qkv.float(), q_ref) < OUT_TOL[dtype]
    for sl in (slice(0, 1), slice(M - 1, M), slice(M // 2, M // 2 + 1)):
        assert rel_l2(qkv[sl].float(), q_ref[sl]) < 2 * OUT_TOL[dtype]
    for c in (0, 7, 319, 320, 639, 640, 959):
        assert rel_l2(qkv[:, c].float(), q_ref[:, c]) < 2 * OUT_TOL[dtype]
    # the launches it replaces
    ln = dict(gamma=gamma, beta=beta, eps=1e-5)
    if pe_on:
        ln.update(pe=pe, rows_per_frame=rpf, pe_frames=frames)
    y3, n3 = ops.gemm(a_ref, wi.to(dtype).contiguous(), bias=bi, residual=res, out_f32=True, ln=ln)
    q3 = ops.gemm(n3, wqkv.to(dtype).contiguous())
    assert rel_l2(y, y3) < (ACC_TOL if not gn else 3e-4) and rel_l2(qkv.float(), q3.float()) < OUT_TOL[dtype]


def test_block_head_fused_rejects_what_it_does_not_cover(dev):
    """The C entry point validates its own arguments (MIMO_EINVAL -> MimoHipError): another width, both or neither operand,
    an image / frame size that is not a whole number of 128-row panels, a misaligned QKV row stride."""
    import ctypes
    from mimo_amd import lib as L
    C, M = 320, 1024
    dt = torch.float16
    ws = torch.zeros((4 * C, C), device=dev, dtype=dt)
    a = torch.zeros((M, C), device=dev, dtype=dt)
    x = torch.zeros((M, C), device=dev)
    ab = torch.zeros((M // 512, 2, C), device=dev)
    g = torch.ones((C,), device=dev)
    y = torch.zeros((M, C), device=dev)
    q = torch.zeros((M, 3 * C), device=dev, dtype=dt)
    st = torch.cuda.current_stream().cuda_stream

    def call(A=None, X=None, AB=None, rpi=0, c=C, ldq=3 * C, pe=None, rpf=0, frames=0):
        return L.call("mimo_block_head_fused", 0, None if A is None else A.data_ptr(), C, None if X is None else X.data_ptr(), C,
                      None if AB is None else AB.data_ptr(), rpi, ws.data_ptr(), g.data_ptr(), None, 0, g.data_ptr(), g.data_ptr(), 1e-5,
                      None if pe is None else pe.data_ptr(), rpf, frames, y.data_ptr(), C, q.data_ptr(), ldq, M, c, st)

    call(A=a)                                  # fine
    call(X=x, AB=ab, rpi=512)                  # fine
    for bad in (dict(A=a, c=640), dict(A=a, X=x, AB=ab, rpi=512), dict(), dict(X=x, AB=ab, rpi=100), dict(X=x, rpi=512),
                dict(A=a, ldq=3 * C + 4), dict(A=a, pe=g.repeat(4).view(4, C), rpf=100, frames=4)):
        with pytest.raises(L.MimoHipError):
            call(**bad)
    torch.cuda.synchronize()


def test_block_head_fused_is_batch_invariant(dev):
    """A row's y / qkv bits depend on that row, its image's affine, its frame's table row and the weights only: a CFG half
    (b = 1) alone equals the same rows of the b = 2 launch bit for bit."""
    from mimo_amd import ops
    from mimo_amd.packing import pack_block_head_stream
    dtype, C, rpi, F_ = torch.float16, 320, 1024, 3
    M = 2 * F_ * rpi
    ws = pack_block_head_stream(rnd((C, C), dev, torch.float32, 10, C ** -0.5), rnd((3 * C, C), dev, torch.float32, 12, C ** -0.5), dtype)
    vec = [rnd((C,), dev, torch.float32, s, 0.1) for s in (11, 13, 14)]
    x = rnd((M, C), dev, torch.float32, 1)
    ab = rnd((M // rpi, 2, C), dev, torch.float32, 2)
    pe = rnd((F_, C), dev, torch.float32, 3)
    run = lambda sl, isl: ops.block_head_fused(ws, vec[0], 1 + vec[1], vec[2], 1e-5, x=x[sl], gn_ab=ab[isl].contiguous(), rows_per_img=rpi,
                                               pe=pe, rows_per_frame=rpi, pe_frames=F_)
    yf, qf = run(slice(0, M), slice(0, 2 * F_))
    for h in range(2):
        yh, qh = run(slice(h * F_ * rpi, (h + 1) * F_ * rpi), slice(h * F_, (h + 1) * F_))
        assert torch.equal(yf[h * F_ * rpi:(h + 1) * F_ * rpi], yh) and torch.equal(qf[h * F_ * rpi:(h + 1) * F_ * rpi], qh)


def test_block_tail_fused_is_batch_invariant(dev):
    """A sharded unit (one CFG half: b = 1) and the full b = 2 launch of the same window must produce the same bits: the
    kernel's result for a row depends on that row, the weights and its image's vector only — not on the panel it falls into
    or on the number of rows in the launch."""
    from mimo_amd import ops
    from mimo_amd.packing import pack_block_tail_stream, pack_ff2_kperm, pack_geglu
    dtype, C, rpi = torch.float16, 320, 4096 + 64   # (an image boundary in the middle of a 128-row panel of the full launch)
    M = 2 * rpi
    o = rnd((M, C), dev, dtype, 1)
    t, x = rnd((M, C), dev, torch.float32, 12), rnd((M, C), dev, torch.float32, 9)
    ib = rnd((2, C), dev, torch.float32, 15)
    w1p, b1p = pack_geglu(rnd((8 * C, C), dev, torch.float32, 2, C ** -0.5), rnd((8 * C,), dev, torch.float32, 3, 0.1), dtype)
    ws = pack_block_tail_stream(rnd((C, C), dev, torch.float32, 10, C ** -0.5), w1p, rnd((C, C), dev, torch.float32, 7, C ** -0.5), dtype)
    w2k = pack_ff2_kperm(rnd((C, 4 * C), dev, torch.float32, 4, (4 * C) ** -0.5), dtype)
    vec = [rnd((C,), dev, torch.float32, s, 0.1) for s in (11, 13, 14, 5, 8)]
    run = lambda sl, ibs: ops.block_tail_fused(o[sl], ws, vec[0], t[sl], 1 + vec[1], vec[2], 1e-5, b1p, w2k, vec[3], vec[4], x[sl],
                                               img_bias=ibs, rows_per_img=rpi)
    full = run(slice(0, M), ib)
    assert torch.equal(full[:rpi], run(slice(0, rpi), ib[:1])) and torch.equal(full[rpi:], run(slice(rpi, M), ib[1:]))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,B,spike", [(64, 1, 0), (100, 3, 0), (1024, 2, 0), (333, 2, 5), (4096, 1, 0)])
def test_attention_single_head_d512(dev, dtype, N, B, spike):
    """The VAE mid-block attention: ONE head of d = 512 over the N tokens of an image (attn512_kernel: head dimension split
    over the four waves of a block, partial scores exchanged through LDS).  Ragged N (not a multiple of the 32-query block
    or the 64-key tile), several images per launch, and a late key spike that moves the running max in a later tile."""
    from mimo_amd import ops
    C = 512
    qkv = rnd((B, N, 3 * C), dev, dtype, 21, 0.5)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    if spike:
        k[B - 1, N - 9] = (q[B - 1, 7].float() * spike).to(dtype)
    out = ops.attention(q, k, v, 1)
    ref = sdpa_ref(q, k, v, 1)
    assert rel_l2(out.float(), ref) < 2.5 * OUT_TOL[dtype]
    for bi in range(B):  # every image separately: a batch-stride mistake would hide in the aggregate
        assert rel_l2(out[bi].float(), ref[bi]) < 2.5 * OUT_TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("d,N,Nb", [(40, 256, 256), (40, 130, 70), (80, 100, 100), (160, 64, 0)])
def test_attention_fp8_qk_variant(dev, dtype, d, N, Nb):
    """mimo_attention_fp8qk (BASELINE configs[4], opt-in): Q.K^T on the e4m3 MFMA.  Pinned against a torch reference whose
    Q and K are rounded to float8_e4m3fn first (then fp32 softmax / P.V): the kernel must reproduce THAT to the usual one
    output rounding; the accuracy cost of the 8-bit operands against the 16-bit kernel is reported, not gated."""
    from mimo_amd import ops
    heads, B = 4, 4
    C = heads * d
    qkv = rnd((B, N, 3 * C), dev, dtype, 31)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    f8 = lambda t: t.float().to(torch.float8_e4m3fn).float()
    kw = {}
    if Nb:
        bank = rnd((Nb, 2 * C), dev, dtype, 32)
        kw = dict(k2=bank[:, :C], v2=bank[:, C:], seg2_first_batch=2)
        ref_u = sdpa_ref(f8(q[:2]), f8(k[:2]), v[:2], heads)
        kc = torch.cat([f8(k[2:]), f8(kw["k2"])[None].expand(2, -1, -1)], dim=1)
        vc = torch.cat([v[2:].float(), kw["v2"].float()[None].expand(2, -1, -1)], dim=1)
        ref = torch.cat([ref_u, sdpa_ref(f8(q[2:]), kc, vc, heads)], dim=0)
    else:
        ref = sdpa_ref(f8(q), f8(k), v, heads)
    with ops.fp8_qk(True):
        out = ops.attention(q, k, v, heads, **kw)
    full = ops.attention(q, k, v, heads, **kw)
    e8, cost = rel_l2(out.float(), ref), rel_l2(out.float(), full.float())
    print(f"fp8 QK^T d={d} N={N}+{Nb} {dtype}: vs fp8-operand reference {e8:.2e}; accuracy cost vs the 16-bit kernel {cost:.2e}")
    assert e8 < 2.5 * OUT_TOL[dtype]
    assert 1e-4 < cost < 0.2  # it IS a different (coarser) computation, and not a broken one


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_forced_rescale(dev, dtype):
    """Spike a late key so the running max jumps in a later KV tile (online-softmax rescale path)."""
    from mimo_amd import ops
    heads, d, N = 2, 40, 192
    C = heads * d
    q = rnd((1, N, C), dev, dtype, 1)
    k = rnd((1, N, C), dev, dtype, 2)
    v = rnd((1, N, C), dev, dtype, 3)
    k[0, 150] = (q[0, 7].float() * 3).to(dtype)  # huge score for query 7 at key 150 (third tile)
    out = ops.attention(q, k, v, heads)
    assert rel_l2(out.float(), sdpa_ref(q, k, v, heads)) < 2.5 * OUT_TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("d,N,Nb,spike", [(40, 256, 256, 0), (40, 130, 70, 0), (40, 192, 0, 4), (40, 200, 64, 4), (40, 192, 0, 40),
                                          (40, 600, 300, 4), (40, 1024, 1024, 0), (40, 64, 0, 0), (40, 1, 3, 0),
                                          (80, 100, 100, 0), (160, 64, 0, 4)])
def test_attention_prescaled_q(dev, dtype, d, N, Nb, spike):
    """q already carries softmax_scale * log2(e) (folded into W_q): C-ABI scale <= 0.  d = 40 runs the variant that
    keeps an integer reference max in a spare MFMA k-slot; the spike forces that reference to move in a late tile,
    and a strongly negative first tile exercises the downward move."""
    from mimo_amd import ops
    heads, B = 4, 4
    C = heads * d
    g = 1.4426950408889634 * d ** -0.5
    qkv = rnd((B, N, 3 * C), dev, dtype, 11)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    if spike:  # spike = 40: a logit of several hundred log2 units, beyond bf16's exactly representable integers
        k[1, N - 20] = (q[1, 7].float() * spike).to(dtype)   # late huge score for query 7 of batch 1
        k[0, :64] = (-q[0, 5].float() * 2).to(dtype)     # whole first tile strongly negative for query 5 of batch 0
    qs = (q.float() * g).to(dtype)                        # what the GEMM with the folded W_q would have produced
    q_eff = qs.float() / g                                # the reference sees exactly the rounded operand
    kw = {}
    if Nb:
        bank = rnd((Nb, 2 * C), dev, dtype, 12)
        kw = dict(k2=bank[:, :C], v2=bank[:, C:], seg2_first_batch=2)
        ref_u = sdpa_ref(q_eff[:2], k[:2], v[:2], heads)
        kc = torch.cat([k[2:], kw["k2"][None].expand(2, -1, -1)], dim=1)
        vc = torch.cat([v[2:], kw["v2"][None].expand(2, -1, -1)], dim=1)
        ref = torch.cat([ref_u, sdpa_ref(q_eff[2:], kc, vc, heads)], dim=0)
    else:
        ref = sdpa_ref(q_eff, k, v, heads)
    out = ops.attention(qs, k, v, heads, q_prescaled=True, **kw)
    assert rel_l2(out.float(), ref) < 2.5 * OUT_TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("d,Fr,HW", [(40, 24, 16), (80, 8, 9), (160, 24, 4), (40, 32, 5), (160, 3, 7), (64, 17, 3), (40, 1, 2),
                                     (80, 24, 33)])
def test_temporal_attention(dev, dtype, d, Fr, HW):
    from mimo_amd import ops
    heads, b = 8, 2
    C = heads * d
    qkv = rnd((b * Fr * HW, 3 * C), dev, dtype, 1)
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    out = ops.temporal_attention(q, k, v, b, Fr, HW, heads)

    def to_seq(t):  # "(b f) d c -> (b d) f c"
        return t.float().reshape(b, Fr, HW, C).permute(0, 2, 1, 3).reshape(b * HW, Fr, C)

    ref = sdpa_ref(to_seq(q), to_seq(k), to_seq(v), heads)
    ref = ref.reshape(b, HW, Fr, C).permute(0, 2, 1, 3).reshape(b * Fr * HW, C)
    assert rel_l2(out.float(), ref) < 2.5 * OUT_TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_softmax_rows(dev, dtype):
    from mimo_amd import ops
    x = rnd((37, 1000), dev, torch.float32, 1) * 3
    out = ops.softmax_rows(x, dtype, scale=0.3)
    assert rel_l2(out.float(), torch.softmax(x * 0.3, -1)) < OUT_TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_layout_roundtrip_and_gather(dev, dtype):
    from mimo_amd import ops
    b, C, Fr, H, W = 2, 4, 6, 5, 3
    x = rnd((b, C, Fr, H, W), dev, torch.float32, 1)
    idx = torch.tensor([4, 5, 0, 1], dtype=torch.int32, device=dev)
    tok = ops.ncfhw_to_tokens(x, dtype, frame_idx=idx, cpad=8)
    assert tok.shape == (b * 4, H, W, 8)
    ref = x[:, :, idx.long()].permute(0, 2, 3, 4, 1).reshape(b * 4, H, W, C).to(dtype)
    assert torch.equal(tok[..., :C], ref) and torch.count_nonzero(tok[..., C:]) == 0
    back = ops.tokens_to_ncfhw(tok, b, C, 4, H, W, scale=2.0)
    assert torch.allclose(back, 2.0 * x[:, :, idx.long()].to(dtype).float())
    img = ops.tokens_to_image(tok.float().contiguous(), b * 4, H, W)
    assert torch.allclose(img, (tok[..., :3].float() / 2 + 0.5).clamp(0, 1).permute(0, 3, 1, 2))


def test_window_accumulate_and_cfg_ddim(dev):
    from mimo_amd import ops
    C, Fr, H, W, Fw = 4, 10, 4, 4, 6
    lat = rnd((1, C, Fr, H, W), dev, torch.float32, 1)
    acc = torch.zeros((2, C, Fr, H, W), device=dev)
    cnt = torch.zeros((Fr,), device=dev)
    acc_ref, cnt_ref = acc.clone(), cnt.clone()
    for s, frames in enumerate([[0, 1, 2, 3, 4, 5], [4, 5, 6, 7, 8, 9], [8, 9, 0, 1, 2, 3]]):
        pred = rnd((2 * Fw, H, W, 4), dev, torch.float32, 10 + s)
        fi = torch.tensor(frames, dtype=torch.int32, device=dev)
        ops.window_accumulate(pred, fi, acc, cnt)
        p5 = pred.reshape(2, Fw, H, W, C).permute(0, 4, 1, 2, 3)
        acc_ref[:, :, fi.long()] += p5
        cnt_ref[fi.long()] += 1
    assert torch.equal(acc, acc_ref) and torch.equal(cnt, cnt_ref)
    g, sa, s1, sap, s1p = 3.5, 0.6, 0.8, 0.9, math.sqrt(1 - 0.81)
    lat2 = lat.clone()
    ops.cfg_ddim_step(acc, cnt, lat2, True, g, sa, s1, sap, s1p)
    npred = acc_ref / cnt_ref[None, None, :, None, None]
    npred = npred[0:1] + g * (npred[1:2] - npred[0:1])
    x0 = sa * lat - s1 * npred
    eps = sa * npred + s1 * lat
    assert torch.allclose(lat2, sap * x0 + s1p * eps, rtol=1e-5, atol=1e-6)
