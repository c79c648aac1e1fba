"""Phase trace of gemm_stream_kernel (tune build): two tracer threads of block 0 — wave 0 (MFMAs first, then DMA issue and
stores) and wave 4 (stores and DMA issue first, then MFMAs) — stamp s_memtime around every phase of a step.
  python tools/stream_trace.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIMO_HIP_LIB", os.path.join(ROOT, "mimo_amd", "libmimo_hip_tune.so"))
os.environ["MIMO_GEMM_TRACE"] = "1"

import numpy as np  # noqa: E402
import torch  # noqa: E402

from mimo_amd import lib as L, ops  # noqa: E402
from mimo_amd.packing import pack_geglu  # noqa: E402

NAMES = {(2, 7): "vmwait", (7, 3): "barrier", (3, 4): "mfma", (4, 5): "-", (5, 9): "dma_issue", (9, 6): "epilogue",
         (3, 10): "epilogue", (10, 9): "dma_issue", (9, 4): "mfma", (5, 6): "-", (6, 2): "-", (8, 2): "-", (6, 8): "a_load"}


def fetch():
    buf = (ctypes.c_ulonglong * 4096)()
    fn = L.load().mimo_tune_trace
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    fn.restype = ctypes.c_int
    assert fn(buf, 4096) == 0
    return np.frombuffer(buf, dtype=np.uint64).copy()


def report(name, fn):
    for _ in range(3):
        fn()
    fetch()
    fn()
    a = fetch()
    for who, sl in (("wave 0 (MFMAs, DMA issue, stores)", a[:2000]), ("wave 4 (stores, DMA issue, MFMAs)", a[2000:4000])):
        ev = [(int(v >> np.uint64(56)), int(v & np.uint64((1 << 56) - 1))) for v in sl if v]
        cyc = [(t, c) for t, c in ev if t not in (0xfe, 0xff)]
        acc, steps = {}, 0
        for (t0, c0), (t1, c1) in zip(cyc, cyc[1:]):
            k = NAMES.get((t0, t1), f"{t0}->{t1}")
            acc[k] = acc.get(k, 0) + c1 - c0
            steps += (t0, t1) == (2, 7)
        tot = cyc[-1][1] - cyc[0][1]
        print(f"{name} | {who}: {tot} cycles, {steps} steps, {tot/steps:.0f} per step: " +
              "  ".join(f"{k} {v/steps:.0f}" for k, v in sorted(acc.items(), key=lambda kv: -kv[1]) if k != "-"), flush=True)


def main():
    dt, dev = torch.float16, torch.device("cuda:0")
    M = 196608
    A = torch.randn(M, 320, device=dev).to(dt)
    W = (torch.randn(960, 320, device=dev) * 0.02).to(dt)
    report("qkv N960", lambda: ops.gemm(A, W))
    wp, bp = pack_geglu(torch.randn(2560, 320, device=dev) * 0.02, torch.zeros(2560, device=dev), dt)
    report("geglu N2560", lambda: ops.gemm(A, wp, bias=bp, geglu=True))


if __name__ == "__main__":
    main()
