"""ff4_kernel (csrc/ff_tail4.hip: 4 waves x 512 registers) against ff_fused_kernel (8 waves x 256 registers) behind the same
entry points, on the tune library (MIMO_FF_TAIL4 = 0 | 1 picks the kernel per launch):
  * outputs (and the 32-row-slab column statistics) compared BIT FOR BIT: mimo_ff_fused and mimo_block_tail_fused, full and ragged
    row counts, per-image vector with panels that straddle two images, fp16 and bf16;
  * launch time of both at the level-0 shape of configs[1] (M = 48 x 4096 rows), inputs rotating through a pool larger than the
    Infinity Cache, three interleaved rounds.
    python tools/ff4_check.py [--time-only] > profiles/r5_ff4_check.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIMO_HIP_LIB", os.path.join(ROOT, "mimo_amd", "libmimo_hip_tune.so"))

import torch  # noqa: E402

from mimo_amd import ops  # noqa: E402
from mimo_amd.packing import pack_block_tail_stream, pack_ff2_kperm, pack_geglu  # noqa: E402

C = 320


def weights(dev, dt, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    w = dict(wo=r(C, C, sc=C ** -0.5), bo=r(C, sc=0.1), gamma=1 + r(C, sc=0.2), beta=r(C, sc=0.2), w1=r(8 * C, C, sc=C ** -0.5),
             b1=r(8 * C, sc=0.1), w2=r(C, 4 * C, sc=(4 * C) ** -0.5), b2=r(C, sc=0.1), wp=r(C, C, sc=C ** -0.5), bp=r(C, sc=0.1))
    w["w1p"], w["b1p"] = pack_geglu(w["w1"], w["b1"], dt)
    w["w2k"] = pack_ff2_kperm(w["w2"], dt)
    w["ws"] = pack_block_tail_stream(w["wo"], w["w1p"], w["wp"], dt)
    return w


def use(k):
    os.environ["MIMO_FF_TAIL4"] = str(k)


def run_ff(w, a, res):
    return ops.ff_fused(a, w["w1p"], w["b1p"], w["w2k"], w["b2"], res)


def run_tail(w, o, t, x, ib=None, rpi=1, colstats=False):
    return ops.block_tail_fused(o, w["ws"], w["bo"], t, w["gamma"], w["beta"], 1e-5, w["b1p"], w["w2k"], w["b2"], w["bp"], x,
                                img_bias=ib, rows_per_img=rpi, colstats=colstats)


def check(dev):
    ok = True
    for dt in (torch.float16, torch.bfloat16):
        w = weights(dev, dt)
        for M, rpi, cs in ((128, 0, 0), (33, 0, 0), (8192 + 77, 1000, 0), (40000, 4096 + 64, 0), (1024, 128, 128), (4096 * 3, 4096, 4096),
                           (196608, 4096, 4096)):
            g = torch.Generator(device="cpu").manual_seed(M)
            a = torch.randn(M, C, generator=g).to(dev).to(dt)
            t = torch.randn(M, C, generator=g).to(dev) + 0.5
            x = torch.randn(M, C, generator=g).to(dev)
            ib = None
            if rpi:
                nimg = (M + rpi - 1) // rpi
                ib = torch.randn(nimg, 3 * C, generator=g).to(dev)[:, C:2 * C]
            outs = {}
            for k in (0, 1):
                use(k)
                f = run_ff(w, a, t)
                y = run_tail(w, a, t, x, ib, rpi or 1, colstats=cs)
                outs[k] = (f, y, ops.stats_of(y) if cs else None)
            torch.cuda.synchronize()
            e_ff = torch.equal(outs[0][0], outs[1][0])
            e_tl = torch.equal(outs[0][1], outs[1][1])
            fin = bool(torch.isfinite(outs[1][1]).all())
            d_ff = float((outs[0][0].float() - outs[1][0].float()).abs().max())
            r_tl = float((outs[0][1] - outs[1][1]).norm() / outs[0][1].norm())
            r_cs = 0.0
            if cs:   # (mean, sum of squared deviations) per 32-row slab and column
                r_cs = max(float((outs[0][2][:, k] - outs[1][2][:, k]).norm() / outs[0][2][:, k].norm()) for k in (0, 1))
            # the shipped ff4_kernel adds the fp32 operands BEHIND the projections (FF4_TRICKLE): y and out differ from ff_fused_kernel's by
            # fp32 rounding order, which flips the half rounding of a few LayerNorm outputs / hidden activations (the same class of
            # difference the unit tests allow against the torch reference); tools/ff4_variants.py checks the exact-order build of
            # the same source (-DFF4_TRICKLE=0) against ff_fused_kernel bit for bit
            good = e_ff and fin and (e_tl or r_tl < (2e-4 if dt == torch.float16 else 4e-4)) and r_cs < 2e-4
            ok &= good
            print(f"{str(dt)[6:]:9s} M {M:7d} rows/img {rpi:5d}  ff_fused bit-equal {e_ff} (max |d| {d_ff:.2e})  block_tail bit-equal {e_tl} "
                  f"(rel-L2 {r_tl:.2e})  colstats rel-L2 {r_cs:.1e}  finite {fin}  {'ok' if good else 'BAD'}", flush=True)
    print("ALL GOOD" if ok else "MISMATCH")
    return ok


def timed(fn, pool, iters=12):
    for i in range(3):
        fn(pool[i % len(pool)])
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for i in range(iters):
        fn(pool[i % len(pool)])
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / iters


def bench(dev):
    dt = torch.float16
    w = weights(dev, dt)
    M, HW = 48 * 4096, 4096
    pool = [(torch.randn(M, C, device=dev).to(dt), torch.randn(M, C, device=dev), torch.randn(M, C, device=dev)) for _ in range(3)]
    ib = torch.randn(48, C, device=dev)
    fl_ff = 2 * M * C * 8 * C + 2 * M * 4 * C * C
    fl_tail = fl_ff + 2 * 2 * M * C * C
    best = {}
    for _ in range(3):
        for k in (0, 1):
            use(k)
            best[("ff_fused", k)] = min(best.get(("ff_fused", k), 1e9), timed(lambda p: run_ff(w, p[0], p[1]), pool))
            best[("block_tail", k)] = min(best.get(("block_tail", k), 1e9), timed(lambda p: run_tail(w, p[0], p[1], p[2], ib, HW, HW), pool))
    print(f"# M = {M}, C = {C}, fp16; ms per launch, best of 3 interleaved rounds, cold inputs")
    for name, fl in (("ff_fused", fl_ff), ("block_tail", fl_tail)):
        for k in (0, 1):
            v = best[(name, k)]
            print(f"{name:11s} {'ff4_kernel (4 waves x 512 regs)' if k else 'ff_fused_kernel (8 waves x 256)':32s} {v:7.3f} ms  {fl / v / 1e9:6.0f} TFLOP/s")


TAGS = {10: "panel start", 11: "to_out tile 0", 12: "to_out tile 1", 13: "to_out tile 2", 14: "to_out tile 3", 15: "to_out tile 4",
        18: "LayerNorm + operand done", 19: "FF position 0", 1: "FF position", 20: "drain", 23: "proj_out tile 0", 24: "proj_out tile 1",
        25: "proj_out tile 2", 26: "proj_out tile 3", 27: "proj_out tile 4", 28: "stores", 30: "kernel end (before drain wait)", 31: "kernel end"}


def trace(dev):
    """tune build: thread 0 of block 0 stamps the cycle counter behind every position's barrier (ff4_kernel MODE 2)"""
    import ctypes
    import numpy as np
    from mimo_amd import lib as L
    dt = torch.float16
    w = weights(dev, dt)
    M, HW = 48 * 4096, 4096
    a = torch.randn(M, C, device=dev).to(dt)
    t = torch.randn(M, C, device=dev)
    x = torch.randn(M, C, device=dev)
    ib = torch.randn(48, C, device=dev)
    use(1)
    fn = L.load().mimo_tune_trace
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    fn.restype = ctypes.c_int
    buf = (ctypes.c_ulonglong * 4096)()
    os.environ["MIMO_FF_TRACE"] = "1"
    for _ in range(3):
        run_tail(w, a, t, x, ib, HW, HW)
        torch.cuda.synchronize()
        assert fn(buf, 4096) == 0
    os.environ["MIMO_FF_TRACE"] = "0"
    v = np.frombuffer(buf, dtype=np.uint64)[:2000]
    ev = [(int(e >> np.uint64(56)), int(e & np.uint64((1 << 56) - 1))) for e in v if e]
    print(f"# ff4_kernel<MODE 2> phase trace of block 0, wave 0 (cycle counter behind each position's barrier), {len(ev)} stamps")
    panel, ff = -1, []
    for (tg, c), (tg2, c2) in zip(ev, ev[1:]):
        if tg == 10:
            panel += 1
            print(f"panel {panel}")
        if tg == 1 or tg == 19:
            ff.append(c2 - c)
            if tg2 == 20:
                arr = np.array(ff)
                print(f"   feed-forward: {len(ff)} positions, {arr.sum()} cycles; per position min {arr.min()} median {int(np.median(arr))} max {arr.max()}"
                      f" (first {ff[0]}, last {ff[-1]})")
                ff = []
            continue
        print(f"   {TAGS.get(tg, tg):32s} {c2 - c:8d} cycles")
    print(f"total {ev[-1][1] - ev[0][1]} cycles")


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    if "--trace" in sys.argv:
        trace(dev)
        sys.exit(0)
    if "--time-only" not in sys.argv:
        check(dev)
    bench(dev)
