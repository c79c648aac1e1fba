"""Phase trace and ablations of ff_fused_kernel (tune build): threads 0 (wave 0: column half 0) and 256 (wave 4: column half 1,
the DMA issuer) of block 0 stamp the cycle counter around every phase of a steady-state step; the ablations time the launch
without DMAs (stale tiles: compute only), without the MFMA phases (stream + barriers only) and with GELU -> identity.
    python tools/ff_trace.py [mode ...]      mode: 1 = mimo_ff_proj_fused, 2 = mimo_block_tail_fused,
                                             3 / 4 = mimo_block_head_fused on a half operand + residual + table / on the fp32 input
                                             (steady-state step = one QKV tile: "ff2" column = its 40 MFMAs, "ff1" = the two stores)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHIPPED = "--shipped" in sys.argv   # time the shipped library's launches only (no trace, no ablations)
if not SHIPPED:
    os.environ.setdefault("MIMO_HIP_LIB", os.path.join(ROOT, "mimo_amd", "libmimo_hip_tune.so"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from mimo_amd import lib as L, ops  # noqa: E402
from mimo_amd.packing import pack_block_head_stream, pack_block_tail_stream, pack_ff2_kperm, pack_geglu, pack_proj_tail  # noqa: E402

NAMES = {(1, 2): "vmwait", (2, 3): "barrier", (3, 4): "dma_issue", (4, 5): "ff2", (5, 6): "ff1", (6, 1): "loop"}


def fetch():
    buf = (ctypes.c_ulonglong * 4096)()
    fn = L.load().mimo_tune_trace
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    fn.restype = ctypes.c_int
    assert fn(buf, 4096) == 0
    return np.frombuffer(buf, dtype=np.uint64).copy()


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def report(name, fn):
    if SHIPPED:
        print(f"{name} | shipped library: {timed(fn):.3f} ms", flush=True)
        return
    os.environ["MIMO_FF_TRACE"] = "1"
    for _ in range(2):
        fn()
    fetch()
    fn()
    a = fetch()
    os.environ["MIMO_FF_TRACE"] = "0"
    for who, sl in (("wave 0 (half 0)", a[:2000]), ("wave 4 (half 1, DMA issuer)", a[2000:4000])):
        ev = [(int(v >> np.uint64(56)), int(v & np.uint64((1 << 56) - 1))) for v in sl if v]
        acc, steps = {}, 0
        for (t0, c0), (t1, c1) in zip(ev, ev[1:]):
            if (t0, t1) not in NAMES or c1 < c0:
                continue  # panel boundary
            k = NAMES[(t0, t1)]
            acc[k] = acc.get(k, 0) + c1 - c0
            steps += (t0, t1) == (1, 2)
        tot = sum(acc.values())
        print(f"{name} | {who}: {steps} steps, {tot/steps:.0f} cycles per step: " +
              "  ".join(f"{k} {acc.get(k, 0)/steps:.0f}" for k in ("vmwait", "barrier", "dma_issue", "ff2", "ff1", "loop")), flush=True)
    # the panel's prologue (block_tail_fused) and tail: cycles between consecutive stamps, averaged over the block's panels
    PRO = {(10, 11): "loads issued", (11, 12): "Wo tile 0", (12, 13): "Wo tile 1", (13, 14): "Wo tile 2", (14, 15): "Wo tile 3",
           (15, 16): "Wo tile 4", (16, 17): "LN statistics", (17, 18): "operand exchange", (18, 19): "first FF step",
           (6, 20): "drain wait", (19, 20): "drain wait", (20, 21): "drain FF2 + exchange", (21, 22): "x load + 5 Wp tiles",
           (10, 16): "operand load + 5 projection tiles", (16, 18): "y store + LN + operand exchange", (6, 22): "(end of panel)"}
    for who, sl in (("wave 0", a[:2000]), ("wave 4", a[2000:4000])):
        ev = [(int(v >> np.uint64(56)), int(v & np.uint64((1 << 56) - 1))) for v in sl if v]
        acc, cnt = {}, {}
        for (t0, c0), (t1, c1) in zip(ev, ev[1:]):
            if (t0, t1) in PRO and c1 >= c0:
                acc[PRO[(t0, t1)]] = acc.get(PRO[(t0, t1)], 0) + c1 - c0
                cnt[PRO[(t0, t1)]] = cnt.get(PRO[(t0, t1)], 0) + 1
        if acc:
            print(f"{name} | {who} per panel: " + "  ".join(f"{k} {acc[k]/cnt[k]:.0f}" for k in dict.fromkeys(PRO.values()) if k in acc), flush=True)
    for abl, what in ((0, "full"), (1, "no DMAs (compute on stale tiles)"), (2, "no MFMA phases (stream + barriers)"), (3, "GELU -> identity")):
        os.environ["MIMO_FF_ABLATE"] = str(abl)
        print(f"{name} | {what}: {timed(fn):.3f} ms", flush=True)
    os.environ["MIMO_FF_ABLATE"] = "0"


def main(modes):
    dt, dev = torch.float16, torch.device("cuda:0")
    M, C = 196608, 320
    g = torch.Generator(device="cpu").manual_seed(1)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    a, res, x = r(M, C).to(dt), r(M, C), r(M, C)
    w1p, b1p = pack_geglu(r(8 * C, C, sc=C ** -0.5), r(8 * C, sc=0.1), dt)
    w2, wp, wo = r(C, 4 * C, sc=(4 * C) ** -0.5), r(C, C, sc=C ** -0.5), r(C, C, sc=C ** -0.5)
    w2k, wpk = pack_ff2_kperm(w2, dt), pack_proj_tail(wp, dt)
    b2, bp, bo, gm, bt = r(C, sc=0.1), r(C, sc=0.1), r(C, sc=0.1), 1 + r(C, sc=0.1), r(C, sc=0.1)
    ws = pack_block_tail_stream(wo, w1p, wp, dt)
    ib = r(48, C)
    if 1 in modes:
        report("ff_proj_fused", lambda: ops.ff_proj_fused(a, w1p, b1p, w2k, b2, res, wpk, bp, x))
    if 2 in modes:
        report("block_tail_fused", lambda: ops.block_tail_fused(a, ws, bo, res, gm, bt, 1e-5, b1p, w2k, b2, bp, x, img_bias=ib, rows_per_img=4096))
        base = lambda: ops.gemm(a, wo.to(dt), bias=bo, img_bias=ib, rows_per_img=4096, residual=res, out_f32=True, ln=dict(gamma=gm, beta=bt))
        print(f"the to_out + LayerNorm launch it absorbs: {timed(base):.3f} ms", flush=True)
        pair = lambda: (base(), ops.ff_proj_fused(a, w1p, b1p, w2k, b2, res, wpk, bp, x))
        print(f"to_out + LayerNorm launch followed by ff_proj_fused: {timed(pair):.3f} ms", flush=True)
    if 3 in modes or 4 in modes:
        wh = pack_block_head_stream(wo, r(3 * C, C, sc=C ** -0.5), dt)
        pe = r(24, C, sc=0.5)
        if 3 in modes:
            report("block_head_fused (half operand + residual + table)",
                   lambda: ops.block_head_fused(wh, bo, gm, bt, 1e-5, a=a, residual=res, pe=pe, rows_per_frame=4096, pe_frames=24))
        if 4 in modes:
            ab = r(48, 2, C)
            report("block_head_fused (fp32 input + GroupNorm affine)",
                   lambda: ops.block_head_fused(wh, bo, gm, bt, 1e-5, x=x, gn_ab=ab, rows_per_img=4096))


if __name__ == "__main__":
    main([int(v) for v in sys.argv[1:] if v.isdigit()] or [1, 2])
