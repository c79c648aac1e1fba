"""How much of the in-situ / isolated gap of the weight-heavy launches is cold weights?  The same launch timed with
  warm   the same weight tensor every time (the isolated microbenchmark),
  cold   a different weight tensor per launch out of a pool larger than the 256 MB last-level cache (what a forward sees:
         every weight is read once per forward, 2.6 GB in total),
  touch  cold, but a plain read of that weight tensor runs in front of the launch (untimed): the weights then sit in L2 / MALL.
Events bracket the launch only.
    python tools/cold_weights.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimo_amd import ops  # noqa: E402
from mimo_amd.packing import pack_conv  # noqa: E402


def timed(fn_for, pool, pre=None, iters=40):
    ev = []
    for i in range(iters + 5):
        w = pool[i % len(pool)]
        if pre is not None:
            pre(w)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn_for(w)
        e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev[5:])
    return t[len(t) // 2] * 1e3


def main():
    dev, dt = torch.device("cuda:0"), torch.float16
    sink = torch.zeros(1, device=dev)
    touch = lambda w: sink.add_(w.view(-1)[::64].float().sum())   # one element per 128-byte line
    flush = torch.empty(600 * 1024 * 1024 // 4, device=dev)

    def run(name, make_w, fn, w_bytes):
        n_pool = max(2, int(700e6 // w_bytes) + 1)
        pool = [make_w() for _ in range(n_pool)]
        warm = timed(fn, pool[:1])
        cold = timed(fn, pool)
        tch = timed(fn, pool, pre=touch)
        print(f"{name:44s} W {w_bytes/1e6:6.1f} MB  warm {warm:7.1f} us  cold {cold:7.1f} us  cold+touch {tch:7.1f} us", flush=True)
        del pool

    for (M, N, K, res) in [(3072, 1280, 1280, True), (3072, 3840, 1280, False), (12288, 1280, 1280, True), (12288, 3840, 1280, False),
                           (12288, 1280, 5120, True), (49152, 1920, 640, False), (49152, 640, 640, True), (49152, 640, 2560, True)]:
        A = torch.randn(M, K, device=dev).to(dt)
        R = torch.randn(M, N, device=dev) if res else None
        mk = lambda: (torch.randn(N, K, device=dev) * 0.02).to(dt)
        run(f"gemm M{M} N{N} K{K}{' +res32' if res else ''}", mk,
            lambda w: ops.gemm(A, w, residual=R, out_f32=res), N * K * 2)
    for (hw, cin, cout) in [(8, 1280, 1280), (16, 1280, 1280), (32, 640, 640), (8, 2560, 1280)]:
        x = torch.randn(48, hw, hw, cin, device=dev).to(dt)
        mk = lambda: pack_conv(torch.randn(cout, cin, 3, 3, device=dev) * 0.02, dt)
        run(f"conv3x3 {hw}x{hw} {cin}->{cout}", mk, lambda w: ops.conv2d(x, w, cout, out_f32=True), cout * cin * 9 * 2)


if __name__ == "__main__":
    main()
