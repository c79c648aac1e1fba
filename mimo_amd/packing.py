"""Weight re-layouts from the reference checkpoint layout (torch nn.Module state-dict tensors)
to what the gfx950 kernels consume.  Done once per loaded state dict, on the device."""
import torch


def pack_conv(weight, dtype, shortcut=None, cin_pad=None, cout_pad=None):
    """nn.Conv2d weight [Cout, Cin, kh, kw] -> [Cout(+pad), kh*kw*Cin(+pad) (+ Cin2)], K index = (ky*kw + kx)*Cin + ci.

    shortcut: optional 1x1 conv weight [Cout, Cin2, 1, 1] appended as an extra tap (fused ResBlock shortcut).
    cin_pad / cout_pad: zero-pad channels (thin first/last convs: Cin must be a multiple of 8, Cout of 4)."""
    co, ci, kh, kw = weight.shape
    w = weight.detach().float().permute(0, 2, 3, 1)  # [co, kh, kw, ci]
    if cin_pad is not None and cin_pad > ci:
        w = torch.nn.functional.pad(w, (0, cin_pad - ci))
    w = w.reshape(co, -1)
    if shortcut is not None:
        w = torch.cat([w, shortcut.detach().float().reshape(co, -1)], dim=1)
    if cout_pad is not None and cout_pad > co:
        w = torch.nn.functional.pad(w, (0, 0, 0, cout_pad - co))
    return w.to(dtype).contiguous()


def split_hi_lo(w, dtype):
    """fp32 tensor -> (hi, lo) in `dtype` with hi = round(w), lo = round(w - hi): w = hi + lo to ~22 mantissa bits."""
    w = w.detach().float()
    hi = w.to(dtype)
    return hi, (w - hi.float()).to(dtype)


def pack_conv_split3(weight, dtype, shortcut=None, cin_pad=None, cout_pad=None, k_pad=None):
    """pack_conv for the split-operand policy (ops.split3): every tap's Cin block becomes [Whi | Wlo | Whi] (3 Cin columns),
    matching an input whose channels are [hi | hi | lo]; the fused 1x1 shortcut segment likewise.  cin_pad pads Cin BEFORE the
    tripling (the input's channel count), k_pad the tripled per-tap block (a thin first conv: 3 * 8 -> 32 channels)."""
    co, ci, kh, kw = weight.shape
    w = weight.detach().float().permute(0, 2, 3, 1)  # [co, kh, kw, ci]
    if cin_pad is not None and cin_pad > ci:
        w = torch.nn.functional.pad(w, (0, cin_pad - ci))
    hi, lo = split_hi_lo(w, dtype)
    w3 = torch.cat([hi, lo, hi], dim=-1)
    if k_pad is not None and k_pad > w3.shape[-1]:
        w3 = torch.nn.functional.pad(w3, (0, k_pad - w3.shape[-1]))
    w3 = w3.reshape(co, -1)
    if shortcut is not None:
        sh, sl = split_hi_lo(shortcut.detach().float().reshape(co, -1), dtype)
        w3 = torch.cat([w3, sh, sl, sh], dim=1)
    if cout_pad is not None and cout_pad > co:
        w3 = torch.nn.functional.pad(w3, (0, 0, 0, cout_pad - co))
    return w3.contiguous()


def pack_linear_split3(weight, dtype, rows_pad=None):
    """Linear weight [N, K] -> [N (+pad), 3K] = [Whi | Wlo | Whi] for an A operand made by ops.split3."""
    hi, lo = split_hi_lo(weight.reshape(weight.shape[0], -1), dtype)
    w3 = torch.cat([hi, lo, hi], dim=1)
    if rows_pad is not None and rows_pad > w3.shape[0]:
        w3 = torch.nn.functional.pad(w3, (0, 0, 0, rows_pad - w3.shape[0]))
    return w3.contiguous()


def pack_linear_split2(weight, dtype):
    """Linear weight [N, K] -> [N, 2K] = [Whi | Wlo] for a half A operand repeated twice along K ([a | a]): only the weight
    is split (the operand is already a 16-bit tensor, e.g. an attention output)."""
    hi, lo = split_hi_lo(weight.reshape(weight.shape[0], -1), dtype)
    return torch.cat([hi, lo], dim=1).contiguous()


def pack_conv_taps(weight, dtype, cout_pad=None):
    """nn.Conv2d 3x3 weight [Cout, Cin, 3, 3] -> [9 * Cout(+pad), Cin], row (3 ky + kx) * Cout + c: the weight of the GEMM half
    of a thin-output convolution (ops.conv3x3_thin_out: every pixel's contribution to the nine outputs around it)."""
    co, ci, kh, kw = weight.shape
    assert kh == 3 and kw == 3
    w = weight.detach().float()
    if cout_pad is not None and cout_pad > co:
        w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 0, 0, cout_pad - co))
    return w.permute(2, 3, 0, 1).reshape(9 * w.shape[0], ci).to(dtype).contiguous()


def pad_vec(v, n):
    v = v.detach().float()
    if v.numel() < n:
        v = torch.nn.functional.pad(v, (0, n - v.numel()))
    return v.contiguous()


def pack_geglu(weight, bias, dtype):
    """GEGLU proj Linear(dim, 2*inner): rows [0, inner) = value, [inner, 2*inner) = gate
    (diffusers GEGLU: hidden, gate = proj(x).chunk(2)).  Interleave in blocks of 16 rows
    [16 value | 16 gate] so one wave holds matching value/gate MFMA tiles."""
    two_inner, dim = weight.shape
    inner = two_inner // 2
    assert inner % 16 == 0
    w = weight.detach().float()
    wv = w[:inner].reshape(inner // 16, 16, dim)
    wg = w[inner:].reshape(inner // 16, 16, dim)
    wp = torch.stack([wv, wg], dim=1).reshape(two_inner, dim)
    b = bias.detach().float()
    bp = torch.stack([b[:inner].reshape(-1, 16), b[inner:].reshape(-1, 16)], dim=1).reshape(two_inner)
    return wp.to(dtype).contiguous(), bp.contiguous()


def pack_ln_fold(weight, gamma, beta, bias, dtype):
    """LayerNorm folded into the projection that consumes it (mimo_epilogue_ext a_row_stats / a_colsum; C = 640 / 1280):
        LayerNorm(x) @ W^T + b = rstd * (x @ (gamma o W)^T - mean * colsum) + (W @ beta + b)
    weight: fp32-convertible [N, K] in the row order the consuming kernel expects (for GEGLU: pack_geglu's fp32 output and its
    packed bias).  Returns dict(w = (gamma o W) in `dtype`, colsum = fp32 [N] row sums of that ROUNDED weight — the centring is
    then exact for what the MFMAs sum —, bias = fp32 [N] = W @ beta + b)."""
    w = weight.detach().double()
    g, be = gamma.detach().double(), beta.detach().double()
    wf = (w * g[None, :]).float().to(dtype).contiguous()
    c2 = w @ be
    if bias is not None:
        c2 = c2 + bias.detach().double()
    return dict(w=wf, colsum=wf.double().sum(1).float().contiguous(), bias=c2.float().contiguous())


def ff2_kperm():
    """Position p = 8 g + j of a 32-wide K block holds original index 4 g + j (j < 4) | 16 + 4 g + (j - 4) (j >= 4): the order
    in which the lanes of an MFMA result tile pair hold a 32-column chunk (lane group g: columns 4g..4g+3 of each 16-tile)."""
    return [4 * (p // 8) + (p % 8) if (p % 8) < 4 else 16 + 4 * (p // 8) + (p % 8) - 4 for p in range(32)]


def pack_ff2_kperm(weight, dtype):
    """FeedForward net.2 Linear(4C, C) weight [C, 4C] for mimo_ff_fused: the K axis permuted inside every 32-block so that
    the GEGLU chunk can feed the second MFMA straight from the first one's accumulator registers."""
    co, k = weight.shape
    assert k % 32 == 0
    idx = torch.tensor(ff2_kperm(), device=weight.device)
    w = weight.detach().float().reshape(co, k // 32, 32)[:, :, idx].reshape(co, k)
    return w.to(dtype).contiguous()


def pack_proj_tail(weight, dtype):
    """proj_out weight [C, C] (Linear, or a 1x1 conv reshaped) for mimo_ff_proj_fused: rows in tile order — tile q (64 rows) =
    output columns 32q..32q+31 followed by 160+32q..160+32q+31, the two column halves the kernel's wave pairs own — and the
    K axis permuted inside every 32-block like pack_ff2_kperm (its operand comes straight from accumulator registers)."""
    co, k = weight.shape
    assert co % 64 == 0 and co // 2 % 32 == 0 and k % 32 == 0
    half = co // 2
    rows = []
    for q in range(co // 64):
        rows += list(range(32 * q, 32 * q + 32)) + list(range(half + 32 * q, half + 32 * q + 32))
    ridx = torch.tensor(rows, device=weight.device)
    kidx = torch.tensor(ff2_kperm(), device=weight.device)
    w = weight.detach().float()[ridx].reshape(co, k // 32, 32)[:, :, kidx].reshape(co, k)
    return w.to(dtype).contiguous()


def pack_rows_tail(weight, dtype):
    """A [C, C] Linear weight whose operand arrives in natural K order (the attention to_out of mimo_block_tail_fused): rows in
    the tile order of pack_proj_tail, K axis untouched."""
    co, k = weight.shape
    assert co % 64 == 0 and co // 2 % 32 == 0
    half = co // 2
    rows = []
    for q in range(co // 64):
        rows += list(range(32 * q, 32 * q + 32)) + list(range(half + 32 * q, half + 32 * q + 32))
    return weight.detach().float()[torch.tensor(rows, device=weight.device)].to(dtype).contiguous()


def pack_block_tail_stream(to_out_w, ff1_w_packed, proj_out_w, dtype):
    """The weight stream of mimo_block_tail_fused, one [10 C, C] tensor in the order the kernel walks it: to_out (rows in tile
    order) | the GEGLU-packed FF1 weight (pack_geglu) with its K axis permuted — its operand, the LayerNorm output, comes
    straight from accumulator registers — | proj_out (pack_proj_tail)."""
    return torch.cat([pack_rows_tail(to_out_w, dtype), pack_ff2_kperm(ff1_w_packed, dtype), pack_proj_tail(proj_out_w, dtype)], 0).contiguous()


def pack_block_head_stream(proj_w, qkv_w, dtype):
    """The weight stream of mimo_block_head_fused, one [4 C, C] tensor in the order the kernel walks it: the leading projection
    (proj_in | an attention's to_out; rows in tile order, K natural: its operand comes from memory) | the fused [Wq; Wk; Wv]
    (natural row order = output column order, K axis permuted: its operand, the LayerNorm output, comes straight from
    accumulator registers)."""
    assert proj_w.shape[0] == proj_w.shape[1] and qkv_w.shape == (3 * proj_w.shape[0], proj_w.shape[1])
    return torch.cat([pack_rows_tail(proj_w, dtype), pack_ff2_kperm(qkv_w, dtype)], 0).contiguous()
