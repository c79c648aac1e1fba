"""Small-M dense GEMMs of the 8 x 8 level (M = 3072): 128-row tiles (MIMO_GEMM_BM64=0) against 64-row tiles (=1), tune library.
Interleaved timing, bit-identity of the two forms.  GPU box:  python tools/small_m_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIMO_HIP_LIB", os.path.join(ROOT, "mimo_amd", "libmimo_hip_tune.so"))
from mimo_amd import lib as L, ops  # noqa: E402
from mimo_amd.packing import pack_ln_fold  # noqa: E402


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    dt = torch.float16
    g = torch.Generator(device="cpu").manual_seed(0)
    print(f"{'case':44s} {'BM128 us':>9s} {'BM64 us':>9s}  bit-identical")
    for M in (3072, 1536, 6144):
        for (N, K) in ((1280, 1280), (1280, 5120), (640, 640)):
            A = torch.randn(M, K, generator=g).to(dev).to(dt)
            W = (torch.randn(N, K, generator=g) * 0.02).to(dev).to(dt)
            R = torch.randn(M, N, generator=g).to(dev)
            b = torch.randn(N, generator=g).to(dev)
            gam, bet = torch.randn(N, generator=g).to(dev), torch.randn(N, generator=g).to(dev)
            cases = [("half out", lambda: ops.gemm(A, W, bias=b)),
                     ("f32 + residual", lambda: ops.gemm(A, W, bias=b, residual=R, out_f32=True))]
            if ops.ln_foldable(N, M):
                cases.append(("f32 + residual, rows (LN-fold producer)",
                              lambda: ops.gemm(A, W, bias=b, residual=R, out_f32=True, ln=dict(gamma=gam, beta=bet, fold=True))))
            for name, fn in cases:
                res, tms = [], []
                for v in ("0", "1"):
                    os.environ["MIMO_GEMM_BM64"] = v
                    L.call("mimo_reload_tuning")
                    out = fn()
                    res.append(out)
                best = [1e9, 1e9]
                for _ in range(3):
                    for i, v in enumerate(("0", "1")):
                        os.environ["MIMO_GEMM_BM64"] = v
                        L.call("mimo_reload_tuning")
                        with ops.split_k(False):
                            best[i] = min(best[i], timeit(fn))

                def flat(o):
                    if isinstance(o, tuple):
                        return [o[0], o[1].xh, o[1].stats]
                    return [o]
                same = all(torch.equal(x, y) for x, y in zip(flat(res[0]), flat(res[1])))
                print(f"M{M} N{N} K{K} {name:28s} {best[0]:9.1f} {best[1]:9.1f}  {same}", flush=True)


if __name__ == "__main__":
    main()
