"""Cycle-counter trace of ONE block of the gemm_kernel family (tune build of the library, MIMO_GEMM_TRACE=1):
thread 0 of block 0 stamps s_memtime at fixed points of the K loop and the epilogue (gemm_conv.hip MIMO_TRACE).
Prints, per shape, where that block's cycles went:
  vmwait   s_waitcnt on the K-tile's DMAs            barrier   waiting for the other waves
  body     DMA issue + fragment reads + MFMA issue   epilogue  loop end -> stores retired
Usage on the GPU box:  python tools/gemm_trace.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIMO_HIP_LIB", os.path.join(ROOT, "mimo_amd", "libmimo_hip_tune.so"))
os.environ["MIMO_GEMM_TRACE"] = "1"

import numpy as np  # noqa: E402
import torch  # noqa: E402

from mimo_amd import lib as L  # noqa: E402
from mimo_amd import ops  # noqa: E402
from mimo_amd.packing import pack_conv, pack_geglu  # noqa: E402


def fetch():
    buf = (ctypes.c_ulonglong * 4096)()
    fn = L.load().mimo_tune_trace
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    fn.restype = ctypes.c_int
    rc = fn(buf, 4096)
    assert rc == 0, rc
    a = np.frombuffer(buf, dtype=np.uint64)
    a = a[a != 0]
    return [(int(v >> np.uint64(56)), int(v & np.uint64((1 << 56) - 1))) for v in a]


def report(name, fn, flops):
    for _ in range(3):
        fn()
    fetch()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    fn()
    en.record()
    torch.cuda.synchronize()
    ms = st.elapsed_time(en)
    tr = fetch()
    if not tr:
        print(f"{name}: launch {ms*1e3:.0f} us ({flops/ms/1e9:.0f} TF/s); no trace (kernel without trace points)")
        return
    real = [t for tag, t in tr if tag in (0xfe, 0xff)]
    cyc = [(tag, t) for tag, t in tr if tag not in (0xfe, 0xff)]
    t0, t1 = cyc[0][1], cyc[-1][1]
    ghz = (t1 - t0) / ((real[-1] - real[0]) * 10.0) if len(real) >= 2 and real[-1] > real[0] else float("nan")
    acc = {"vmwait": 0, "barrier": 0, "body": 0, "epilogue": 0, "other": 0}
    ntile = nk = 0
    prev_tag, prev_t = cyc[0]
    for tag, t in cyc[1:]:
        d = t - prev_t
        if prev_tag == 2 and tag == 7:
            acc["vmwait"] += d
        elif prev_tag == 7 and tag == 3:
            acc["barrier"] += d
        elif prev_tag == 3 and tag == 4:
            acc["body"] += d
            nk += 1
        elif prev_tag == 5 and tag == 6:
            acc["epilogue"] += d
            ntile += 1
        else:
            acc["other"] += d
        prev_tag, prev_t = tag, t
    tot = t1 - t0
    print(f"{name}: launch {ms*1e3:.0f} us ({flops/ms/1e9:.0f} TF/s); block 0: {tot} cycles = {tot/ghz/1e3 if ghz == ghz else float('nan'):.0f} us "
          f"at {ghz:.2f} GHz; tiles {ntile}, K-tiles {nk}")
    print("    " + "  ".join(f"{k} {v} ({v/tot*100:.0f}%)" for k, v in acc.items()) +
          (f"  | per K-tile: vmwait {acc['vmwait']/nk:.0f} barrier {acc['barrier']/nk:.0f} body {acc['body']/nk:.0f}" if nk else "") +
          (f"  | epilogue/tile {acc['epilogue']/ntile:.0f}" if ntile else ""), flush=True)


def main():
    dt = torch.float16
    dev = torch.device("cuda:0")
    n = 48
    print(f"# {torch.cuda.get_device_name(0)}; s_memtime cycles of thread 0 / block 0; clock = cycles / s_memrealtime (100 MHz)")
    for (hw, cin, cout) in [(64, 320, 320), (32, 640, 640), (16, 1280, 1280), (64, 960, 320)]:
        x = torch.randn(n, hw, hw, cin, device=dev).to(dt)
        w = pack_conv(torch.randn(cout, cin, 3, 3, device=dev) * 0.02, dt)
        b = torch.zeros(cout, device=dev)
        report(f"conv3x3 {hw}x{hw} {cin}->{cout}", lambda: ops.conv2d(x, w, cout, bias=b, out_f32=True), 2 * n * hw * hw * cout * 9 * cin)
    for (M, N, K) in [(196608, 320, 320), (196608, 960, 320), (196608, 320, 1280), (49152, 1920, 640), (12288, 1280, 5120)]:
        A = torch.randn(M, K, device=dev).to(dt)
        W = (torch.randn(N, K, device=dev) * 0.02).to(dt)
        R = torch.randn(M, N, device=dev)
        report(f"gemm M{M} N{N} K{K}", lambda: ops.gemm(A, W), 2 * M * N * K)
        if N == 320:
            report(f"gemm+res32 M{M} N{N} K{K}", lambda: ops.gemm(A, W, residual=R, out_f32=True), 2 * M * N * K)
    # the LayerNorm-fused C x C linears of level 0 (BM = 128 persistent kernel, whole rows per tile)
    M, C = 196608, 320
    A = torch.randn(M, C, device=dev).to(dt)
    W = (torch.randn(C, C, device=dev) * 0.05).to(dt)
    R = torch.randn(M, C, device=dev)
    b = torch.zeros(C, device=dev)
    ln = dict(gamma=torch.ones(C, device=dev), beta=torch.zeros(C, device=dev), eps=1e-5)
    report("gemm+ln M196608 N320 K320 (proj_in)", lambda: ops.gemm(A, W, bias=b, out_f32=True, ln=ln), 2 * M * C * C)
    report("gemm+res32+ln M196608 N320 K320 (to_out)", lambda: ops.gemm(A, W, bias=b, residual=R, out_f32=True, ln=ln), 2 * M * C * C)
    for (M, dim) in [(196608, 320), (49152, 640), (12288, 1280)]:
        A = torch.randn(M, dim, device=dev).to(dt)
        wp, bp = pack_geglu(torch.randn(8 * dim, dim, device=dev) * 0.02, torch.zeros(8 * dim, device=dev), dt)
        report(f"geglu-ff1 M{M} dim{dim}", lambda: ops.gemm(A, wp, bias=bp, geglu=True), 2 * M * 8 * dim * dim)


if __name__ == "__main__":
    main()
