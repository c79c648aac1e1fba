"""Tensor-level wrappers over the C-ABI (mimo_amd.lib): torch is used only to own device
memory and the stream; every op below is ONE call into libmimo_hip.so.

Activation convention: channels-last token-major.  An image batch is a contiguous tensor
[n, H, W, C] (equivalently [n*H*W, C]); the residual stream is fp32, MFMA operands are
fp16/bf16 ("half16").
"""
import ctypes
import threading

import torch

from . import lib as L

_DT = {torch.float16: L.F16, torch.bfloat16: L.BF16}

# Optional accounting of the algorithmic FLOPs actually launched (2*MACs of every GEMM / conv / attention
# product); bench.py sets COUNTER = {"flops": 0, "launches": 0} around a forward to price the roofline.
COUNTER = None


# Optional per-launch HIP-event brackets of the MFMA kernels (bench.py roofline of the dominant kernel family):
# EVENTS = [] makes gemm()/conv2d()/attention() record (family, start event, end event, flops) on the launch stream.
EVENTS = None
TAGS = None   # with EVENTS: a parallel list of shape tags (tools/forward_bound.py's per-shape table)


def _count(flops):
    if COUNTER is not None:
        COUNTER["flops"] += flops
        COUNTER["launches"] += 1


def _nbytes(*tensors):
    """Algorithmic HBM bytes of a launch: every operand read once, every result written once."""
    return sum(t.numel() * t.element_size() for t in tensors if t is not None)


class _Bracket:
    def __init__(self, family, flops, nbytes=0, tag=""):
        self.family, self.flops, self.nbytes, self.tag = family, flops, nbytes, tag

    def __enter__(self):
        if EVENTS is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *a):
        if EVENTS is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            EVENTS.append((self.family, self.e0, e1, self.flops, self.nbytes))
            if TAGS is not None:
                TAGS.append(self.tag)
        return False


def dt_code(dtype):
    try:
        return _DT[dtype]
    except KeyError:
        raise L.MimoHipError(f"compute dtype must be float16 or bfloat16, got {dtype}")


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


# Split-K changes the fp32 summation order of a long reduction as a function of the row count M, i.e. of the batch
# size.  Runs that must be bit-identical across batch decompositions (one long clip sharded over ranks vs the same clip
# on one GPU) switch it off: `with ops.split_k(False): ...` adds MIMO_EPI_NO_SPLITK to every gemm / conv2d call made
# inside the block (a per-call flag of the C-ABI; the library itself keeps no state).
_SPLIT_K = True


class split_k:
    def __init__(self, enabled):
        self.enabled = bool(enabled)

    def __enter__(self):
        global _SPLIT_K
        self.saved, _SPLIT_K = _SPLIT_K, self.enabled
        return self

    def __exit__(self, *a):
        global _SPLIT_K
        _SPLIT_K = self.saved
        return False


def split_k_enabled():
    return _SPLIT_K


_WS = {}


_TLS = threading.local()  # workspace slot / base of the calling host thread (two threads may drive two pipelines)


class workspace_slot:
    """Launches issued inside `with workspace_slot(i)` use split-K scratch buffer i of their device.  The pipeline gives each
    of its concurrent window streams its own slot: two split-K launch pairs on different streams must not share partials."""

    def __init__(self, slot):
        self.slot = slot

    def __enter__(self):
        self.prev, _TLS.slot = getattr(_TLS, "slot", 0), self.slot

    def __exit__(self, *a):
        _TLS.slot = self.prev
        return False


class workspace_base:
    """Shifts every workspace slot chosen inside the block by `base`: two pipelines that run concurrently on different streams
    (bench.py --clips-in-flight 2) each use their own set of split-K scratch buffers."""

    def __init__(self, base):
        self.base = base

    def __enter__(self):
        self.prev, _TLS.base = getattr(_TLS, "base", 0), self.base

    def __exit__(self, *a):
        _TLS.base = self.prev
        return False


def _workspace(device):
    """Split-K scratch handed to mimo_gemm / mimo_conv2d (include/mimo_hip.h, workspace convention): ONE fp32 buffer
    per (device, workspace slot).  Only the split-K partial-sum launch pair of one call touches it, and launches that
    share a slot are issued on one stream at a time (during hipGraph capture: the capture stream), so stream order makes
    sharing safe; concurrent streams take different slots (workspace_slot)."""
    dev = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    key = (dev, getattr(_TLS, "base", 0) + getattr(_TLS, "slot", 0))
    ws = _WS.get(key)
    if ws is None:
        ws = _WS[key] = torch.empty(L.load().mimo_workspace_bytes() // 4, device=device, dtype=torch.float32)
    return ws


# GroupNorm column statistics (mimo_epilogue_ext.colstats): a producer called with colstats=<rows per image> attaches
# the fp32 [M/32, 2, N] tensor to its output as `_cs`; group_norm() consumes it instead of re-reading the tensor.
# The decision depends on the IMAGE size only, never on the batch: a frame's statistics must be computed the same way
# however many frames share the launch (the sharded long-clip mode reproduces the single-GPU bits).  Images below
# COLSTATS_MIN_HW pixels keep split-K available instead (the 8x8 level: one more pass over 16 MB is the cheaper way);
# above COLSTATS_MAX_HW the statistics epilogue costs the producer more than the separate statistics pass it saves
# (measured at 64x64 x 320: +34 us on the convolution vs 25 us for the pass, which reads the just-written tensor from
# the Infinity Cache; at 32x32 x 640 +9 us vs 36 us, profiles/r2_microbench_fused.txt).
COLSTATS_MIN_HW = 256
COLSTATS_MAX_HW = 1024


def _want_colstats(rows_per_image, M):
    hw = int(rows_per_image or 0)
    return COLSTATS_MIN_HW <= hw <= COLSTATS_MAX_HW and hw % 32 == 0 and M % hw == 0


def with_stats(t, cs):
    """Re-attach column statistics after a view (views are new tensor objects)."""
    if cs is not None:
        t._cs = cs
    return t


def stats_of(t):
    return getattr(t, "_cs", None)


def _ext(colstats=None, ln_out=None, ln=None, row_half=None, row_stats=None, a_fold=None, colsum=None):
    e = L.EpilogueExt()
    e.row_half, e.row_stats = _ptr(row_half), _ptr(row_stats)
    if a_fold is not None:
        e.a_row_stats, e.a_colsum, e.a_slots, e.a_eps = a_fold.stats.data_ptr(), colsum.data_ptr(), a_fold.stats.shape[1], a_fold.eps
    e.colstats = _ptr(colstats)
    e.ln_out = _ptr(ln_out)
    if ln is not None:
        e.ln_gamma, e.ln_beta, e.ln_pe = ln["gamma"].data_ptr(), ln["beta"].data_ptr(), _ptr(ln.get("pe"))
        e.ln_eps = float(ln.get("eps", 1e-5))
        e.ln_pe_frames = int(ln.get("pe_frames", 0) or 0)
        e.ln_rows_per_frame = int(ln.get("rows_per_frame", 0) or 0)
    return e


def _chk(t, name):
    if not t.is_cuda:
        raise L.MimoHipError(f"{name} must be a device tensor: mimo_amd has no CPU path")


def _is_f32(t):
    if t.dtype == torch.float32:
        return 1
    if t.dtype in _DT:
        return 0
    raise L.MimoHipError(f"unsupported tensor dtype {t.dtype}")


# The LayerNorm after an N = 640 projection in the GEMM's epilogue (gemm_ln640_kernel: 64 x 640 whole-row tiles).  OFF: measured
# SLOWER than GEMM + LayerNorm launch (166.6 vs 125.3 us at M49152 K640, 391.6 vs 222.5 at K2560, profiles/r5_ln640_bench.txt) —
# one block per CU, 20 serial K-tile steps and a 246 KB store epilogue that overlaps with nothing (the LDS-DMA path itself is not the
# limit: 55-64 B/clk per CU, profiles/r5_lds_fill_probe.txt).  Kept for the record and tested; levels 1-3 use LN_FOLD below instead.
LN_OUT_640 = False


# LayerNorm folded into the projection that consumes it, where no tile holds a whole row (C = 640 / 1280: levels 1-3): the
# producer GEMM leaves a half copy of its rows and per-row partial sums (mimo_epilogue_ext row_half / row_stats), the consumer
# multiplies the raw rows by gamma o W and applies mean / rstd in its epilogue.  The normalised tensor is never written and
# the fp32 tensor never re-read: the layer_norm launch and 4 of its 6 bytes per element disappear.  A property of the layer
# width only (never of the batch).  Numerics: the rounded tensor is the LayerNorm's input instead of its output — priced on
# the oracle as class LNRAW (oracle/error_budget.py): 2.25e-4 against 2.62e-4 for LN at step 3 of configs[0].
LN_FOLD = True
LN_FOLD_MIN_C = 640


class LnFold:
    """The un-normalised operand of a folded LayerNorm: half rows + per-row (sum, sum of squares) partials, eps."""

    def __init__(self, xh, stats, eps):
        self.xh, self.stats, self.eps = xh, stats, float(eps)

    @property
    def shape(self):
        return self.xh.shape


_SLOTS = {}


def row_stat_slots(N):
    if N not in _SLOTS:
        _SLOTS[N] = L.call_int("mimo_row_stat_slots", N)
    return _SLOTS[N]


def ln_foldable(N, M=0):
    """The folded LayerNorm covers width N (mimo_row_stat_slots: whole widest tiles, at most 20 slots) and, if given, row count M
    (the library addresses row_half with 32-bit byte offsets); callers fall back to out + layer_norm otherwise."""
    return LN_FOLD and N >= LN_FOLD_MIN_C and row_stat_slots(N) > 0 and M < (2 ** 31 - 1) // (2 * N)


def gemm(a, w, *, bias=None, img_bias=None, rows_per_img=0, residual=None, out_f32=False, silu=False,
         geglu=False, out=None, out_scale=1.0, colstats=False, ln=None, colsum=None):
    """out[M, N] = epilogue(a[M, K] @ w[N, K]^T); a may be a row-strided view (last stride 1).

    colstats=<rows per image>: also emit GroupNorm column statistics of `out` (attached as out._cs) when the image size
    allows (see COLSTATS_MIN_HW).
    ln=dict(gamma, beta[, eps, pe, rows_per_frame, pe_frames]): also return LayerNorm(out) (+ pe) as a half tensor —
    fused into the epilogue when N == 320 or 640, otherwise a separate mimo_layer_norm launch.  Returns (out, ln_out).
    ln=dict(..., fold=True) with ln_foldable(N): nothing is normalised here; returns (out, LnFold) for a consumer called as
    gemm(LnFold, packing.pack_ln_fold(...)["w"], bias=[...]["bias"], colsum=[...]["colsum"], ...) (a positional table, if any,
    is the consumer's per-image bias)."""
    a_fold = None
    if isinstance(a, LnFold):
        assert colsum is not None and colsum.dtype == torch.float32 and colsum.shape == (w.shape[0],)
        a_fold, a = a, a.xh
    _chk(a, "a")
    assert a.dim() == 2 and a.stride(1) == 1 and w.dim() == 2 and w.is_contiguous()
    assert a.dtype == w.dtype
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K
    n_out = N // 2 if geglu else N
    if out is None:
        out = torch.empty((M, n_out), device=a.device, dtype=torch.float32 if out_f32 else a.dtype)
    assert out.stride(1) == 1
    flags = (L.EPI_SILU if silu else 0) | (L.EPI_GEGLU if geglu else 0) | \
            (L.EPI_OUT_F32 if out.dtype == torch.float32 else 0)
    ldr = 0
    if residual is not None:
        assert residual.dim() == 2 and residual.stride(1) == 1 and residual.shape[0] == M
        flags |= L.EPI_RES_F32 if residual.dtype == torch.float32 else 0
        ldr = residual.stride(0)
    ldib = 0
    if img_bias is not None:
        assert img_bias.dim() == 2 and img_bias.stride(1) == 1 and img_bias.dtype == torch.float32
        ldib = img_bias.stride(0)
    if not _SPLIT_K:
        flags |= L.EPI_NO_SPLITK
    # N = 320: 128-row whole-row tiles of the persistent kernel; N = 640 (round 5, LN_OUT_640): 64-row tiles of gemm_ln640_kernel
    fuse_ln = (ln is not None and (N == 320 or (N == 640 and LN_OUT_640 and K % 32 == 0)) and not geglu and not silu
               and out.stride(0) == N and out.is_contiguous() and (residual is None or ldr == N)
               and (ln.get("pe") is None or (ln.get("rows_per_frame", 0) % 128 == 0 and ln.get("pe_frames", 0) > 0)))
    fold = (ln is not None and ln.get("fold") and not fuse_ln and ln_foldable(N, M) and out.dtype == torch.float32 and not geglu
            and not silu and ln.get("pe") is None)
    cs = ln_out = row_half = row_stats = None
    if fold:
        row_half = torch.empty((M, N), device=a.device, dtype=a.dtype)
        row_stats = torch.empty((M, row_stat_slots(N), 2), device=a.device, dtype=torch.float32)
    if not fuse_ln and not fold and a_fold is None and not geglu and N % 4 == 0 and _want_colstats(colstats, M):
        cs = torch.empty((M // 32, 2, N), device=a.device, dtype=torch.float32)
    if fuse_ln:
        ln_out = torch.empty((M, N), device=a.device, dtype=a.dtype)
    _count(2 * M * N * K)
    with _Bracket("gemm_kernel", 2 * M * N * K, M * K * a.element_size() + _nbytes(w, out, residual, ln_out, cs, row_half, row_stats)
                  + (a_fold.stats.numel() * 4 if a_fold is not None else 0),
                  f"gemm M{M} N{N} K{K}{' geglu' if geglu else ''}{' ln' if fuse_ln else ''}{' f32' if out_f32 else ''}"
                  f"{' rows' if fold else ''}{' folded' if a_fold is not None else ''}"):
        ws = _workspace(a.device)
        if fold or a_fold is not None:
            ext = _ext(row_half=row_half, row_stats=row_stats, a_fold=a_fold, colsum=colsum)
            L.call("mimo_gemm_ext", dt_code(a.dtype), a.data_ptr(), a.stride(0), w.data_ptr(), out.data_ptr(),
                   out.stride(0), M, N, K, _ptr(bias), _ptr(img_bias), ldib, rows_per_img, _ptr(residual), ldr,
                   float(out_scale), flags, ws.data_ptr(), ws.numel() * 4, ctypes.byref(ext), _stream())
        elif cs is None and ln_out is None:
            L.call("mimo_gemm", dt_code(a.dtype), a.data_ptr(), a.stride(0), w.data_ptr(), out.data_ptr(),
                   out.stride(0), M, N, K, _ptr(bias), _ptr(img_bias), ldib, rows_per_img, _ptr(residual), ldr,
                   float(out_scale), flags, ws.data_ptr(), ws.numel() * 4, _stream())
        else:
            ext = _ext(colstats=cs, ln_out=ln_out, ln=ln if fuse_ln else None)
            L.call("mimo_gemm_ext", dt_code(a.dtype), a.data_ptr(), a.stride(0), w.data_ptr(), out.data_ptr(),
                   out.stride(0), M, N, K, _ptr(bias), _ptr(img_bias), ldib, rows_per_img, _ptr(residual), ldr,
                   float(out_scale), flags, ws.data_ptr(), ws.numel() * 4, ctypes.byref(ext), _stream())
    with_stats(out, cs)
    if ln is None:
        return out
    if fold:
        return out, LnFold(row_half, row_stats, ln.get("eps", 1e-5))
    if ln_out is None:
        ln_out = layer_norm(out, ln["gamma"], ln["beta"], eps=ln.get("eps", 1e-5), dtype=a.dtype, pe=ln.get("pe"),
                            rows_per_frame=ln.get("rows_per_frame", 0), pe_frames=ln.get("pe_frames", 0))
    return out, ln_out


# The fused feed-forward kernel exists for the 320-wide level only (its K = 320 operand lives in registers).  Like every
# kernel choice of this package the decision depends on the layer's width and on a fixed row threshold, never on how many
# images share the launch beyond it (FF_FUSED_MIN_ROWS: below it the panel count cannot fill the chip).
FF_FUSED = True
FF_PROJ_FUSED = True      # ... and the block's output projection + residual folded into the same launch (mimo_ff_proj_fused)
BLOCK_TAIL_FUSED = True   # ... and the attention output projection + the LayerNorm in front (mimo_block_tail_fused)
FF_FUSED_DIM = 320
FF_FUSED_MIN_ROWS = 8192
FF_FUSED_MAX_ROWS = (2 ** 31 - 1) // (4 * 320) - 128   # (M - 1) * ld + C must stay below 2^31 bytes for the fp32 operands (ld = 320)


def ff_fused(a, w1p, b1p, w2k, b2, residual):
    """half [M, C] = residual (fp32) + GEGLU(a @ w1p^T + b1p) @ w2^T + b2 in ONE launch (C = 320): w1p / b1p GEGLU-packed
    (packing.pack_geglu), w2k = packing.pack_ff2_kperm(net.2.weight).  The [M, 4C] intermediate never reaches HBM."""
    _chk(a, "a")
    M, C = a.shape
    assert a.stride(1) == 1 and w1p.shape == (8 * C, C) and w2k.shape == (C, 4 * C) and w1p.is_contiguous() and w2k.is_contiguous()
    assert residual.dtype == torch.float32 and residual.shape == (M, C) and residual.stride(1) == 1
    out = torch.empty((M, C), device=a.device, dtype=a.dtype)
    fl = 2 * M * C * (8 * C + 4 * C)
    _count(fl)
    with _Bracket("gemm_kernel", fl, _nbytes(a, w1p, w2k, residual, out), f"ff_fused M{M}"):
        L.call("mimo_ff_fused", dt_code(a.dtype), a.data_ptr(), a.stride(0), w1p.data_ptr(), _ptr(b1p), w2k.data_ptr(), _ptr(b2),
               residual.data_ptr(), residual.stride(0), out.data_ptr(), out.stride(0), M, C, _stream())
    return out


FF_COLSTATS = True  # the fused tails also emit the GroupNorm column statistics of their output (no statistics pass after them)


def _tail_colstats(colstats, M, C, device):
    """[M/32, 2, C] statistics buffer of a fused tail, or None.  colstats = rows per image; like _want_colstats a function
    of the image size only (whole 32-row slabs per image), never of the batch."""
    hw = int(colstats or 0)
    if FF_COLSTATS and hw >= 32 and hw % 32 == 0 and M % hw == 0:
        return torch.empty((M // 32, 2, C), device=device, dtype=torch.float32)
    return None


def ff_proj_fused(a, w1p, b1p, w2k, b2, residual, wpk, bp, x, colstats=False):
    """fp32 [M, C] = x + (residual + FF(a)) @ Wp^T + bp in ONE launch (C = 320): ff_fused with the block's output projection
    and its residual folded in; wpk = packing.pack_proj_tail(proj_out.weight).  colstats=<rows per image>: the launch also
    emits the GroupNorm column statistics of its output (attached as out._cs)."""
    _chk(a, "a")
    M, C = a.shape
    assert a.stride(1) == 1 and w1p.shape == (8 * C, C) and w2k.shape == (C, 4 * C) and wpk.shape == (C, C)
    assert w1p.is_contiguous() and w2k.is_contiguous() and wpk.is_contiguous()
    for r in (residual, x):
        assert r.dtype == torch.float32 and r.shape == (M, C) and r.stride(1) == 1
    out = torch.empty((M, C), device=a.device, dtype=torch.float32)
    cs = _tail_colstats(colstats, M, C, a.device)
    fl = 2 * M * C * (8 * C + 4 * C + C)
    _count(fl)
    with _Bracket("gemm_kernel", fl, _nbytes(a, w1p, w2k, wpk, residual, x, out, cs), f"ff_proj_fused M{M}"):
        L.call("mimo_ff_proj_fused", dt_code(a.dtype), a.data_ptr(), a.stride(0), w1p.data_ptr(), _ptr(b1p), w2k.data_ptr(),
               _ptr(b2), residual.data_ptr(), residual.stride(0), wpk.data_ptr(), _ptr(bp), x.data_ptr(), x.stride(0),
               out.data_ptr(), out.stride(0), M, C, _ptr(cs), _stream())
    return with_stats(out, cs)


def block_tail_fused(o, wstream, bo, residual, ln_gamma, ln_beta, ln_eps, b1p, w2k, b2, bp, x, img_bias=None, rows_per_img=1,
                     colstats=False):
    """fp32 [M, C] = x + (y + FF(LayerNorm(y))) @ Wp^T + bp with y = residual + o @ Wo^T + bo (+ img_bias per image) in ONE
    launch (C = 320): everything a transformer block does after its attention core plus the owning transformer's proj_out;
    wstream = packing.pack_block_tail_stream(to_out.weight, GEGLU-packed FF1, proj_out.weight)."""
    _chk(o, "o")
    M, C = o.shape
    assert o.stride(1) == 1 and wstream.shape == (10 * C, C) and w2k.shape == (C, 4 * C) and wstream.is_contiguous() and w2k.is_contiguous()
    for r in (residual, x):
        assert r.dtype == torch.float32 and r.shape == (M, C) and r.stride(1) == 1
    ldib = 0
    if img_bias is not None:
        assert img_bias.dim() == 2 and img_bias.stride(1) == 1 and img_bias.dtype == torch.float32 and img_bias.shape[1] == C
        assert img_bias.shape[0] * rows_per_img >= M
        ldib = img_bias.stride(0)
    out = torch.empty((M, C), device=o.device, dtype=torch.float32)
    cs = _tail_colstats(colstats, M, C, o.device)  # colstats=<rows per image>: GroupNorm column statistics of out as out._cs
    fl = 2 * M * C * (C + 8 * C + 4 * C + C)
    _count(fl)
    with _Bracket("gemm_kernel", fl, _nbytes(o, wstream, w2k, residual, x, out, cs), f"block_tail_fused M{M}"):
        L.call("mimo_block_tail_fused", dt_code(o.dtype), o.data_ptr(), o.stride(0), wstream.data_ptr(), _ptr(bo), _ptr(img_bias),
               ldib, rows_per_img, residual.data_ptr(), residual.stride(0), ln_gamma.data_ptr(), ln_beta.data_ptr(), float(ln_eps),
               _ptr(b1p), w2k.data_ptr(), _ptr(b2), _ptr(bp), x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), M, C,
               _ptr(cs), _stream())
    return with_stats(out, cs)


BLOCK_HEAD_FUSED = True   # everything BEFORE the attention core too: (GroupNorm ->) projection -> LayerNorm -> QKV (mimo_block_head_fused)


def block_head_fused(wstream, bi, ln_gamma, ln_beta, ln_eps, *, a=None, x=None, gn_ab=None, rows_per_img=0, residual=None,
                     pe=None, rows_per_frame=0, pe_frames=0):
    """(y fp32 [M, C], qkv half [M, 3C]) in ONE launch (C = 320): y = (residual +) A' @ Wi^T + bi, qkv = (LayerNorm(y) (+ pe))
    @ Wqkv^T with A' = `a` (half, an attention output) or half(x * ga + gb) — `x` the fp32 block input, gn_ab =
    group_norm_affine(...) of the GroupNorm in front of proj_in.  wstream = packing.pack_block_head_stream(Wi, [Wq; Wk; Wv]).
    Neither the normalised input nor the LayerNorm output reaches memory."""
    src = a if a is not None else x
    _chk(src, "a" if a is not None else "x")
    assert (a is None) != (x is None)
    M, C = src.shape
    assert src.stride(1) == 1 and wstream.shape == (4 * C, C) and wstream.is_contiguous()
    dtype = wstream.dtype
    if a is not None:
        assert a.dtype == dtype
    else:
        assert x.dtype == torch.float32 and gn_ab is not None and gn_ab.dtype == torch.float32 and gn_ab.is_contiguous()
        assert gn_ab.shape == (M // rows_per_img, 2, C) and M % rows_per_img == 0 and rows_per_img >= 128
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.shape == (M, C) and residual.stride(1) == 1
    if pe is not None:
        assert pe.dtype == torch.float32 and pe.is_contiguous() and pe.shape[1] == C and pe.shape[0] >= pe_frames > 0
        assert rows_per_frame >= 128
    y = torch.empty((M, C), device=src.device, dtype=torch.float32)
    qkv = torch.empty((M, 3 * C), device=src.device, dtype=dtype)
    fl = 2 * M * C * (C + 3 * C)
    _count(fl)
    with _Bracket("gemm_kernel", fl, _nbytes(src, wstream, residual, y, qkv), f"block_head_fused M{M}{' gn' if x is not None else ''}"):
        L.call("mimo_block_head_fused", dt_code(dtype), _ptr(a), 0 if a is None else a.stride(0), _ptr(x), 0 if x is None else x.stride(0),
               _ptr(gn_ab), int(rows_per_img), wstream.data_ptr(), _ptr(bi), _ptr(residual), 0 if residual is None else residual.stride(0),
               ln_gamma.data_ptr(), ln_beta.data_ptr(), float(ln_eps), _ptr(pe), int(rows_per_frame), int(pe_frames),
               y.data_ptr(), y.stride(0), qkv.data_ptr(), qkv.stride(0), M, C, _stream())
    return y, qkv


def conv2d(x, w, cout, *, ksize=3, stride=1, pad=None, out_hw=None, upsample_to=None, x2=None, bias=None,
           img_bias=None, imgs_per_bias_row=1, residual=None, out_f32=False, silu=False, out_scale=1.0, colstats=False,
           out=None):
    """Channels-last implicit-GEMM conv.  x: half [n, H, W, Cin]; w: packed half [cout, ks*ks*Cin (+Cin2)].

    pad = (pad_top, pad_left); default (ks//2, ks//2).  out_hw defaults to the torch formula for
    symmetric padding.  upsample_to = (Hup, Wup) applies nearest-neighbour upsampling first.
    x2: optional half [n, Hout, Wout, Cin2] fused as an extra 1x1 tap (ResBlock shortcut).
    img_bias: fp32 [n / imgs_per_bias_row, cout] view (row-strided allowed): per-image additive vector.
    """
    _chk(x, "x")
    assert x.dim() == 4 and x.is_contiguous() and w.is_contiguous()
    n, H, W, cin = x.shape
    if pad is None:
        pad = (ksize // 2, ksize // 2)
    Hv, Wv = (upsample_to if upsample_to is not None else (H, W))
    if out_hw is None:
        out_hw = ((Hv + 2 * pad[0] - ksize) // stride + 1, (Wv + 2 * pad[1] - ksize) // stride + 1)
    Ho, Wo = out_hw
    cin2 = 0
    if x2 is not None:
        assert x2.is_contiguous() and x2.shape[:3] == (n, Ho, Wo)
        cin2 = x2.shape[3]
    assert w.shape == (cout, ksize * ksize * cin + cin2), (w.shape, cout, ksize, cin, cin2)
    p = L.ConvParams(n, H, W, cin, Ho, Wo, cout, ksize, stride, pad[0], pad[1],
                     Hv if upsample_to is not None else 0, Wv if upsample_to is not None else 0, cin2,
                     imgs_per_bias_row, 0 if img_bias is None else img_bias.stride(0))
    if img_bias is not None:
        assert img_bias.dim() == 2 and img_bias.stride(1) == 1 and img_bias.dtype == torch.float32
    if out is None:
        out = torch.empty((n, Ho, Wo, cout), device=x.device, dtype=torch.float32 if out_f32 else x.dtype)
    else:  # a caller-owned destination (a row band of a larger tensor)
        assert out.shape == (n, Ho, Wo, cout) and out.is_contiguous() and out.dtype == (torch.float32 if out_f32 else x.dtype)
    flags = (L.EPI_SILU if silu else 0) | (L.EPI_OUT_F32 if out_f32 else 0)
    if residual is not None:
        assert residual.is_contiguous() and residual.shape == out.shape
        flags |= L.EPI_RES_F32 if residual.dtype == torch.float32 else 0
    if not _SPLIT_K:
        flags |= L.EPI_NO_SPLITK
    M = n * Ho * Wo
    cs = None
    if colstats and cout % 4 == 0 and _want_colstats(Ho * Wo, M):
        cs = torch.empty((M // 32, 2, cout), device=x.device, dtype=torch.float32)
    fl = 2 * n * Ho * Wo * cout * (ksize * ksize * cin + cin2)
    _count(fl)
    with _Bracket("gemm_kernel", fl, _nbytes(x, x2, w, out, residual, cs),
                  f"conv{ksize} {Ho}x{Wo} cin{cin}{'+' + str(cin2) if cin2 else ''} cout{cout} s{stride}{' up' if upsample_to else ''}"
                  f"{' f32' if out_f32 else ''} n{n}"):
        ws = _workspace(x.device)
        if cs is None:
            L.call("mimo_conv2d", dt_code(x.dtype), x.data_ptr(), _ptr(x2), w.data_ptr(), out.data_ptr(),
                   ctypes.byref(p), _ptr(bias), _ptr(img_bias), _ptr(residual), float(out_scale), flags,
                   ws.data_ptr(), ws.numel() * 4, _stream())
        else:
            ext = _ext(colstats=cs)
            L.call("mimo_conv2d_ext", dt_code(x.dtype), x.data_ptr(), _ptr(x2), w.data_ptr(), out.data_ptr(),
                   ctypes.byref(p), _ptr(bias), _ptr(img_bias), _ptr(residual), float(out_scale), flags,
                   ws.data_ptr(), ws.numel() * 4, ctypes.byref(ext), _stream())
    return with_stats(out, cs)


def group_norm_stats(x1, *, groups=32, eps=1e-5, x2=None, dtype=None):
    """(mean, rstd) per (image, group) of the virtual channel concat [x1 | x2]: fp32 [n, groups, 2].  From the producers'
    epilogue column statistics when both inputs carry them (no pass over HBM), otherwise one statistics pass."""
    _chk(x1, "x1")
    assert x1.is_contiguous() and (x2 is None or x2.is_contiguous())
    n = x1.shape[0]
    C1 = x1.shape[-1]
    C2 = 0 if x2 is None else x2.shape[-1]
    HW = x1.numel() // (n * C1)
    if dtype is None:
        dtype = x1.dtype if x1.dtype in _DT else torch.float16
    f32 = _is_f32(x1)
    if x2 is not None:
        assert _is_f32(x2) == f32 and x2.numel() // (n * C2) == HW
    cs1, cs2 = stats_of(x1), (None if x2 is None else stats_of(x2))
    stats = torch.empty((n, groups, 2), device=x1.device, dtype=torch.float32)
    def slab_rows(cs):  # pixels per partial of a producer's statistics, or 0 when they do not tile an image
        rows = (n * HW) // cs.shape[0] if (n * HW) % cs.shape[0] == 0 else 0
        return rows if rows and HW % rows == 0 else 0
    if cs1 is not None and (x2 is None or cs2 is not None) and slab_rows(cs1) and (cs2 is None or slab_rows(cs2)):
        # the producers' epilogues already reduced every slab (32 rows: mimo_gemm_ext / mimo_conv2d_ext / the fused tails /
        # mimo_conv3x3_fused with a residual; 256 pixels: mimo_conv3x3_fused tile statistics): merge slabs x group
        # columns, no pass over x
        L.call("mimo_group_norm_stats_slabs", cs1.data_ptr(), C1, slab_rows(cs1), _ptr(cs2), C2, slab_rows(cs2) if cs2 is not None else 0,
               n, HW, groups, float(eps), stats.data_ptr(), _stream())
        return stats
    C = C1 + C2
    if C1 % 4 == 0 and C2 % 4 == 0 and groups <= C // 4 <= 1024 and HW * C >= (1 << 20):
        # row-streaming statistics: one block per (image, pixel slice) reads whole pixel rows; the slices'
        # (S, Q) partials are reduced in fixed order (deterministic)
        # (the slicing is a function of the image size only: a frame's statistics must not depend on how many
        # other frames share the launch, or the sharded long-clip mode would not reproduce the single-GPU bits)
        split = max(1, min(256, HW // 96))
        partials = torch.empty((n * groups * split, 2), device=x1.device, dtype=torch.float32)
    else:
        # one block per (image, group), pixel-sliced for large images
        split = max(1, min(8, HW // 4096))
        partials = torch.empty((n * groups * split, 2), device=x1.device, dtype=torch.float32) if split > 1 else None
    L.call("mimo_group_norm_stats", x1.data_ptr(), C1, _ptr(x2), C2, f32, dt_code(dtype), n, HW, groups,
           float(eps), stats.data_ptr(), _ptr(partials), split, _stream())
    return stats


def group_norm_apply(x1, stats, gamma, beta, *, groups=32, silu=False, x2=None, dtype=None, want_raw=False,
                     want_norm=True):
    """The apply pass of GroupNorm(+SiLU) with GIVEN statistics (fp32 [n, groups, 2], or None with want_norm=False: a
    plain half cast).  x1 / x2 may be row bands [n = 1, rows, W, C] of an image whose statistics were taken over the
    whole image (the band-tiled VAE decode).  Returns (normed_half or None, raw_half or None)."""
    _chk(x1, "x1")
    assert x1.is_contiguous() and (x2 is None or x2.is_contiguous())
    n = x1.shape[0]
    C1 = x1.shape[-1]
    C2 = 0 if x2 is None else x2.shape[-1]
    HW = x1.numel() // (n * C1)
    if dtype is None:
        dtype = x1.dtype if x1.dtype in _DT else torch.float16
    f32 = _is_f32(x1)
    shape = tuple(x1.shape[:-1]) + (C1 + C2,)
    out = torch.empty(shape, device=x1.device, dtype=dtype) if want_norm else None
    raw = torch.empty(shape, device=x1.device, dtype=dtype) if want_raw else None
    if want_norm:
        assert stats is not None and stats.is_contiguous() and stats.shape == (n, groups, 2)
    L.call("mimo_group_norm_apply", x1.data_ptr(), C1, _ptr(x2), C2, f32, dt_code(dtype), n, HW, groups,
           _ptr(stats) if want_norm else None, _ptr(gamma), _ptr(beta), int(silu), _ptr(out), _ptr(raw), _stream())
    return out, raw


def group_norm(x1, gamma, beta, *, groups=32, eps=1e-5, silu=False, x2=None, dtype=None, want_raw=False,
               want_norm=True):
    """GroupNorm(+SiLU) over the virtual channel concat [x1 | x2]; inputs [n, H, W, C*] fp32 or half.

    Returns (normed_half or None, raw_half or None)."""
    stats = group_norm_stats(x1, groups=groups, eps=eps, x2=x2, dtype=dtype) if want_norm else None
    return group_norm_apply(x1, stats, gamma, beta, groups=groups, silu=silu, x2=x2, dtype=dtype, want_raw=want_raw,
                            want_norm=want_norm)


def split3(x, stats=None, gamma=None, beta=None, *, groups=32, silu=False, dtype=torch.float16, ld=None, c_off=0, c_total=0,
           out=None, col=0):
    """The split-operand form of an fp32 activation (mimo_group_norm_apply_split3): half [..., ld >= 3C] = [hi | hi | lo] of
    y = silu?(GroupNorm(x)) with the given statistics (fp32 [n, groups, 2]) or of x itself (stats None) — the A operand of a
    GEMM / convolution whose weight is packing.pack_conv_split3 / pack_linear_split3 ([Whi | Wlo | Whi] along K): both
    operands then carry ~22 mantissa bits through the fp16 MFMAs.  x: fp32 [n, H, W, C] or [M, C]; ld > 3C: zero padding.
    c_total > 0: x is the source that starts at channel c_off of a virtual concat of c_total channels (stats / gamma / beta are
    the concatenated tensor's).  out / col: write the 3C columns at column `col` of an existing [..., ld] tensor (the operands of
    several sources side by side without a concat copy)."""
    _chk(x, "x")
    assert x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] % 8 == 0
    C = x.shape[-1]
    n = x.shape[0] if x.dim() == 4 else 1
    HW = x.numel() // (n * C)
    if out is not None:
        ld = out.shape[-1]
        assert out.is_contiguous() and out.dtype == dtype and out.shape[:-1] == x.shape[:-1] and col % 8 == 0 and col + 3 * C <= ld
    else:
        assert col == 0
        ld = 3 * C if ld is None else int(ld)
        out = (torch.zeros if ld > 3 * C else torch.empty)(tuple(x.shape[:-1]) + (ld,), device=x.device, dtype=dtype)
    if stats is not None:
        ct = c_total if c_total > 0 else C
        assert stats.is_contiguous() and stats.shape == (n, groups, 2) and gamma.numel() == ct and beta.numel() == ct
    L.call("mimo_group_norm_apply_split3", x.data_ptr(), C, dt_code(dtype), n, HW, groups, _ptr(stats), _ptr(gamma), _ptr(beta),
           int(silu), out.data_ptr() + col * out.element_size(), ld, int(c_off), int(c_total), _stream())
    return out


def frames_differ(x):
    """int32 [n] device: 1 where frame i of the contiguous tensor x [n, ...] is not bit-identical to frame i - 1 (entry 0 is 1);
    None when the frame size is not a multiple of 16 bytes."""
    _chk(x, "x")
    assert x.is_contiguous()
    n = x.shape[0]
    fb = x.numel() // n * x.element_size()
    if fb % 16 or x.data_ptr() % 16:
        return None
    d = torch.zeros((n,), device=x.device, dtype=torch.int32)
    L.call("mimo_frames_differ", x.data_ptr(), n, fb, d.data_ptr(), _stream())
    return d


def group_norm_affine(stats, gamma, beta, C, groups=32):
    """GroupNorm folded to a per-(image, channel) affine: fp32 [n, 2, C] with GroupNorm(x)[c] = x * ab[i, 0, c] + ab[i, 1, c]
    (the operand of conv3x3_fused)."""
    _chk(stats, "stats")
    n = stats.shape[0]
    assert stats.shape == (n, groups, 2) and stats.is_contiguous() and gamma.numel() == C and beta.numel() == C
    ab = torch.empty((n, 2, C), device=stats.device, dtype=torch.float32)
    L.call("mimo_group_norm_affine", stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), n, C, groups, ab.data_ptr(), _stream())
    return ab


# Halo-tiled 3x3 convolution that normalises its own input (mimo_conv3x3_fused, csrc/hconv.hip).  Which layers take it
# is a function of the LAYER (image size, channel counts) only, never of the number of images in the launch: a frame's
# bits must not depend on the batch it is computed in.  Below HCONV_MIN_HW pixels per image the 16 x 16 pixel tiles
# quantise worse than the row-tiled kernel's 192-row tiles (32 x 32 x 48 images x 640 channels = 384 blocks on 256 CUs).
HCONV = True
HCONV_MIN_HW = 4096
HCONV_TILE_STATS = True  # conv3x3_fused(tile_stats=True) also emits per-tile GroupNorm statistics of its output
HCONV_SLAB_STATS = True  # ... also with a residual (32-pixel slabs) and for the up-sampling convolution


def hconv_supported(x1, cout, *, x2=None, normed=True, upsample2x=False):
    """True when conv3x3_fused covers this layer (see include/mimo_hip.h: mimo_conv3x3_fused)."""
    if not HCONV or x1.dtype != torch.float32 or x1.dim() != 4 or (x2 is not None and x2.dtype != torch.float32):
        return False
    n, Hs, Ws, C1 = x1.shape
    C2 = 0 if x2 is None else x2.shape[3]
    H, W = (2 * Hs, 2 * Ws) if upsample2x else (Hs, Ws)
    C = C1 + C2
    if H % 16 or W % 16 or H * W < HCONV_MIN_HW or C % 32 or (C2 and (C1 % 64 or C2 % 32)):
        return False
    if cout % 128 and cout % 320:
        return False
    if normed and C > (960 if cout % 320 == 0 else 2560):
        return False
    per_img = max(Hs * Ws * max(C1, C2) * 4, H * W * C * 2, (15 * W + 16) * cout * 4)
    return per_img < 2 ** 31 and Hs * Ws < 2 ** 21


def conv3x3_fused(x1, w, cout, *, x2=None, ab=None, bias=None, img_bias=None, imgs_per_bias_row=1, residual=None,
                  out_scale=1.0, upsample2x=False, want_raw=False, raw_dtype=None, out=None, tile_stats=False):
    """fp32 [n, H, W, cout] = epilogue(conv3x3(silu(x * a + b))) in ONE launch: x = fp32 virtual concat [x1 | x2], ab from
    group_norm_affine (None: plain cast, no SiLU — the up-sampling convolution); w = the packed weight of conv2d
    (columns beyond 9 C, a fused shortcut segment, are ignored).  want_raw: also returns the half cast of x.
    tile_stats: also emit per-(16 x 16 tile | with a residual: 32-pixel slab, channel) statistics of `out` (attached as out._cs): the GroupNorm that
    consumes `out` then merges them instead of making a statistics pass over the tensor."""
    _chk(x1, "x1")
    assert x1.dim() == 4 and x1.is_contiguous() and x1.dtype == torch.float32 and w.is_contiguous()
    n, Hs, Ws, C1 = x1.shape
    C2 = 0
    if x2 is not None:
        assert x2.is_contiguous() and x2.dtype == torch.float32 and x2.shape[:3] == x1.shape[:3]
        C2 = x2.shape[3]
    C = C1 + C2
    H, W = (2 * Hs, 2 * Ws) if upsample2x else (Hs, Ws)
    assert w.shape[0] == cout and w.shape[1] >= 9 * C, (w.shape, cout, C)
    if ab is not None:
        assert ab.shape == (n, 2, C) and ab.is_contiguous() and ab.dtype == torch.float32
    if out is None:
        out = torch.empty((n, H, W, cout), device=x1.device, dtype=torch.float32)
    else:
        assert out.shape == (n, H, W, cout) and out.is_contiguous() and out.dtype == torch.float32
    raw = torch.empty((n, H, W, C), device=x1.device, dtype=raw_dtype or w.dtype) if want_raw else None
    flags = L.EPI_OUT_F32
    if residual is not None:
        assert residual.is_contiguous() and residual.shape == out.shape and residual.dtype == torch.float32
        flags |= L.EPI_RES_F32
    ldib = 0
    if img_bias is not None:
        assert img_bias.dim() == 2 and img_bias.stride(1) == 1 and img_bias.dtype == torch.float32
        ldib = img_bias.stride(0)
    p = L.HconvParams(n, H, W, cout, int(upsample2x), imgs_per_bias_row, ldib)
    ts = None
    if tile_stats and HCONV_TILE_STATS:
        if residual is None and (ab is not None or HCONV_SLAB_STATS):      # per 16 x 16 tile, from the accumulators
            ts = torch.empty((n * (H // 16) * (W // 16), 2, cout), device=x1.device, dtype=torch.float32)
        elif residual is not None and HCONV_SLAB_STATS and ab is not None and not want_raw:
            # per 32-pixel slab (two tile rows), from the stored values (residual included)
            ts = torch.empty((n * H * W // 32, 2, cout), device=x1.device, dtype=torch.float32)
    fl = 2 * n * H * W * cout * 9 * C
    _count(fl)
    with _Bracket("gemm_kernel", fl, _nbytes(x1, x2, out, residual, raw) + cout * 9 * C * w.element_size(),
                  f"hconv {H}x{W} cin{C1}{'+' + str(C2) if C2 else ''} cout{cout}{' up' if upsample2x else ''}"
                  f"{' gn' if ab is not None else ''}{' raw' if want_raw else ''} n{n}"):
        L.call("mimo_conv3x3_fused", dt_code(w.dtype), x1.data_ptr(), C1, _ptr(x2), C2, _ptr(ab), int(ab is not None),
               w.data_ptr(), w.shape[1], out.data_ptr(), ctypes.byref(p), _ptr(bias), _ptr(img_bias), _ptr(residual), _ptr(raw),
               _ptr(ts), float(out_scale), flags, _stream())
    with_stats(out, ts)
    return (out, raw) if want_raw else out


# The EDGES of a UNet take split operands (hi + lo pairs, see split3) under the default policy too: bit 0 the output head
# (conv_norm_out + conv_out: its rounding lands on the prediction undamped), bit 1 the per-clip tables (time-embedding chain,
# collapsed cross-attentions), bit 2 the input convolution (fed fp32 tokens: the latents are not rounded at all).  Three tiny
# layers: +0.3 ms per forward for 15 % of its distance from the fp32 reference (one forward 7.4e-4 -> 6.3e-4, full size 768^2
# 7.3e-4 -> 6.3e-4, configs[0] final 1.13e-3 -> 1.03e-3: tools/edge_split_probe.py, tools/config1_probe.py, round 6).  Bit 3: conv2 and the
# 1x1 shortcut of the LAST TWO resnets of the denoising UNet (ResnetBlock.edge_parts): the level-0 up block in front of the output
# head carries 55 % (configs[0]) to 92 % (512^2 forward) of the error variance of the whole UNet, these two products of its last two
# resnets the cheapest share of it — configs[0] final latents 1.025e-3 -> 9.31e-4 (it was the only full-size case above the 1e-3 bar),
# one forward 6.3e-4 -> 4.9e-4, configs[1] 6.4e-4 -> 5.9e-4, for +2.0 ms per forward = 2.9 % (interleaved A/B; conv1 stays on the fused
# launch: precise.resnet(parts=...); tools/config1_probe.py --scan [--parts], tools/sensitivity_scan.py; profiles/r6_config1_sensitivity_scan*.txt).
# 0 = everything off.
EDGE_SPLIT = 15
THIN_OUT = True  # thin-output 3x3 convolutions (conv_out of the UNet / VAE decoder) as GEMM + tap gather


def conv3x3_thin_out(x, w_taps, cout, *, bias=None, out_scale=1.0):
    """fp32 [n, H, W, cout] = conv3x3(x) (padding 1) for a thin output (cout <= 16): x half [n, H, W, Cin] is read ONCE by a
    GEMM with w_taps = packing.pack_conv_taps(weight) ([9 cout, Cin]); mimo_conv3x3_tapsum gathers the nine contributions of
    every output pixel (the implicit-GEMM kernel would spend a 160-column tile on 4 channels and re-read x per tap)."""
    _chk(x, "x")
    assert x.dim() == 4 and x.is_contiguous() and w_taps.shape == (9 * cout, x.shape[3]) and cout % 4 == 0 and cout <= 16
    n, H, W, C = x.shape
    # (as a 1x1 convolution: mimo_conv2d routes K = 128 | 320 with a thin N to the direct kernel of csrc/thinconv.hip)
    taps = conv2d(x, w_taps, 9 * cout, ksize=1, out_f32=True)
    out = torch.empty((n, H, W, cout), device=x.device, dtype=torch.float32)
    L.call("mimo_conv3x3_tapsum", taps.data_ptr(), 9 * cout, n, H, W, cout, _ptr(bias), out.data_ptr(), float(out_scale), _stream())
    return out


def layer_norm(x, gamma, beta, *, eps=1e-5, dtype=None, pe=None, rows_per_frame=0, pe_frames=0, out_f32=False):
    """LayerNorm over the last dim of x [rows, C] -> half (fp32 if out_f32); optional + pe[(row // rows_per_frame) % pe_frames]."""
    _chk(x, "x")
    assert x.is_contiguous()
    C = x.shape[-1]
    rows = x.numel() // C
    if dtype is None:
        dtype = x.dtype if x.dtype in _DT else torch.float16
    out = torch.empty(x.shape, device=x.device, dtype=torch.float32 if out_f32 else dtype)
    L.call("mimo_layer_norm", x.data_ptr(), _is_f32(x), dt_code(dtype), rows, C, float(eps), gamma.data_ptr(),
           beta.data_ptr(), _ptr(pe), rows_per_frame, pe_frames, None if out_f32 else out.data_ptr(),
           out.data_ptr() if out_f32 else None, _stream())
    return out


# Opt-in fp8 (e4m3) Q.K^T for the spatial attention (BASELINE configs[4]; accuracy reported, not gated).  `with
# ops.fp8_qk(True): ...` routes every attention() call at d in {40, 80, 160} made inside the block to mimo_attention_fp8qk.
_FP8_QK = False


class fp8_qk:
    def __init__(self, enabled):
        self.enabled = bool(enabled)

    def __enter__(self):
        global _FP8_QK
        self.saved, _FP8_QK = _FP8_QK, self.enabled
        return self

    def __exit__(self, *a):
        global _FP8_QK
        _FP8_QK = self.saved
        return False


def attention(q, k, v, heads, *, k2=None, v2=None, seg2_first_batch=0, scale=None, q_prescaled=False):
    """Multi-head attention.  q: [B, Nq, C] view, k/v: [B, Nk, C] views (last stride 1, batch stride =
    N * row stride); optional shared second segment k2/v2: [Nk2, C] views for batches >= seg2_first_batch."""
    _chk(q, "q")
    B, Nq, C = q.shape
    Nk = k.shape[1]
    d = C // heads
    for t, N in ((q, Nq), (k, Nk), (v, Nk)):
        assert t.stride(2) == 1 and (B == 1 or t.stride(0) == N * t.stride(1)), (t.shape, t.stride())
    Nk2 = 0
    ldk2 = ldv2 = 0
    if k2 is not None:
        Nk2 = k2.shape[0]
        assert k2.stride(1) == 1 and v2.stride(1) == 1
        ldk2, ldv2 = k2.stride(0), v2.stride(0)
    out = torch.empty((B, Nq, C), device=q.device, dtype=q.dtype)
    if q_prescaled:
        scale = 0.0  # C-ABI convention: scale <= 0 <=> q already carries softmax_scale * log2(e)
    elif scale is None:
        scale = d ** -0.5
    fl = 4 * Nq * C * (B * Nk + max(B - seg2_first_batch, 0) * Nk2)
    _count(fl)
    with _Bracket("attn_kernel", fl, (B * (Nq * 2 + Nk * 2) * C + (2 * Nk2 * C if Nk2 else 0)) * q.element_size()):
        L.call("mimo_attention_fp8qk" if (_FP8_QK and d in (40, 80, 160)) else "mimo_attention",
               dt_code(q.dtype), q.data_ptr(), q.stride(1), k.data_ptr(), k.stride(1),
               v.data_ptr(), v.stride(1), _ptr(k2), ldk2, _ptr(v2), ldv2, out.data_ptr(), C, B, Nq, Nk, Nk2,
               seg2_first_batch, heads, d, float(scale), _stream())
    return out


def temporal_attention(q, k, v, b, F, HW, heads, *, scale=None):
    """Attention over the F frames of each (batch, pixel, head); q/k/v: [b*F*HW, C] views (last stride 1)."""
    _chk(q, "q")
    C = q.shape[-1]
    d = C // heads
    assert q.shape[0] == b * F * HW and q.stride(1) == 1 and k.stride(1) == 1 and v.stride(1) == 1
    out = torch.empty((q.shape[0], C), device=q.device, dtype=q.dtype)
    if scale is None:
        scale = d ** -0.5
    _count(4 * b * HW * F * F * C)
    L.call("mimo_temporal_attention", dt_code(q.dtype), q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0),
           v.data_ptr(), v.stride(0), out.data_ptr(), C, b, F, HW, heads, d, float(scale), _stream())
    return out


def softmax_rows(x, dtype, scale=1.0, out=None):
    """Row softmax of a (row-strided) fp32 matrix into a (row-strided) half matrix."""
    _chk(x, "x")
    assert x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=dtype)
    assert out.shape == x.shape and out.dtype == dtype and out.stride(1) == 1
    L.call("mimo_softmax_rows", dt_code(dtype), x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0),
           x.shape[0], x.shape[1], float(scale), _stream())
    return out


def ncfhw_to_tokens(x, dtype, *, frame_idx=None, cpad=None, out=None, out_col0=0):
    """[b, C, F, H, W] (fp32/half) -> half tokens [b*F', H, W, cpad]; frame_idx: int32 device tensor."""
    _chk(x, "x")
    assert x.dim() == 5 and x.is_contiguous()
    b, C, F, H, W = x.shape
    Fsel = F if frame_idx is None else frame_idx.numel()
    if cpad is None:
        cpad = C
    if out is None:
        out = torch.empty((b * Fsel, H, W, cpad), device=x.device, dtype=dtype)
    assert out.is_contiguous() and out.dtype == dtype
    L.call("mimo_ncfhw_to_tokens", x.data_ptr(), _is_f32(x), dt_code(dtype), b, C, F, H, W, _ptr(frame_idx), Fsel,
           cpad, out.shape[-1], out_col0, out.data_ptr(), _stream())
    return out


def tokens_to_ncfhw(tok, b, C, F, H, W, *, scale=1.0):
    """tokens [b*F, H, W, ld] -> fp32 [b, C, F, H, W] (channels [0, C))."""
    _chk(tok, "tok")
    assert tok.is_contiguous()
    ld = tok.shape[-1]
    dt = tok.dtype if tok.dtype in _DT else torch.float16
    out = torch.empty((b, C, F, H, W), device=tok.device, dtype=torch.float32)
    L.call("mimo_tokens_to_ncfhw", tok.data_ptr(), _is_f32(tok), dt_code(dt), ld, b, C, F, H, W, float(scale),
           out.data_ptr(), _stream())
    return out


def tokens_to_image(tok, n, H, W):
    """tokens [n, H, W, ld>=3] -> fp32 [n, 3, H, W] = clamp(x/2 + 0.5, 0, 1)."""
    _chk(tok, "tok")
    assert tok.is_contiguous()
    dt = tok.dtype if tok.dtype in _DT else torch.float16
    out = torch.empty((n, 3, H, W), device=tok.device, dtype=torch.float32)
    L.call("mimo_tokens_to_image", tok.data_ptr(), _is_f32(tok), dt_code(dt), tok.shape[-1], n, H, W,
           out.data_ptr(), _stream())
    return out


def cast_half(x, dtype):
    _chk(x, "x")
    assert x.is_contiguous() and x.numel() % 8 == 0
    out = torch.empty(x.shape, device=x.device, dtype=dtype)
    L.call("mimo_cast", x.data_ptr(), _is_f32(x), dt_code(dtype), x.numel(), out.data_ptr(), _stream())
    return out


def window_accumulate(pred_tok, frames, acc, counter):
    """acc[:, :, frames[j]] += pred[:, :, j]; counter[frames[j]] += 1.
    pred_tok: fp32 tokens [bb*Fw, H, W, ld]; acc: fp32 [bb, C, F, H, W]; counter fp32 [F]."""
    _chk(pred_tok, "pred_tok")
    assert pred_tok.dtype == torch.float32 and pred_tok.is_contiguous() and acc.is_contiguous()
    bb, C, F, H, W = acc.shape
    Fw = frames.numel()
    L.call("mimo_window_accumulate", pred_tok.data_ptr(), pred_tok.shape[-1], frames.data_ptr(), Fw, bb, C, F,
           H * W, acc.data_ptr(), counter.data_ptr(), _stream())


def cfg_ddim_step(acc, counter, latents, cfg, guidance, sa, s1, sap, s1p, frames=None):
    """In-place: latents <- DDIM(v-pred) step of CFG-combined, window-averaged prediction; frames (int32 device tensor of
    distinct frame indices): only those frames are advanced (same arithmetic per element)."""
    _chk(latents, "latents")
    assert latents.dtype == torch.float32 and latents.is_contiguous() and acc.is_contiguous()
    _, C, F, H, W = latents.shape
    if frames is not None:
        assert frames.dtype == torch.int32 and frames.is_contiguous() and 0 < frames.numel() <= F
        L.call("mimo_cfg_ddim_step_frames", acc.data_ptr(), counter.data_ptr(), latents.data_ptr(), C, F, H * W, frames.data_ptr(),
               frames.numel(), int(cfg), float(guidance), float(sa), float(s1), float(sap), float(s1p), _stream())
        return
    L.call("mimo_cfg_ddim_step", acc.data_ptr(), counter.data_ptr(), latents.data_ptr(), C, F, H * W, int(cfg),
           float(guidance), float(sa), float(s1), float(sap), float(s1p), _stream())
