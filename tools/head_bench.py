"""mimo_block_head_fused against the launches it replaces, at the level-0 shape of configs[1] (M = 48 x 4096 rows, C = 320):
  gn   : GroupNorm-apply pass + GEMM (proj_in, fused LayerNorm output) + QKV GEMM   vs   one fused head on the fp32 input
  a_res: GEMM (to_out + residual, fused LayerNorm + PE output) + QKV GEMM           vs   one fused head on the half operand
Inputs rotate through a pool larger than the Infinity Cache (cold activations, as inside a forward).
  python tools/head_bench.py > profiles/r5_head_bench.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimo_amd import ops  # noqa: E402
from mimo_amd.packing import pack_block_head_stream  # noqa: E402


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


def main():
    dev = torch.device("cuda:0")
    dt = torch.float16
    C, HW, n = 320, 4096, 48
    M = n * HW
    g = torch.Generator(device="cpu").manual_seed(0)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    wi, wqkv = r(C, C, sc=C ** -0.5), r(3 * C, C, sc=C ** -0.5)
    bi, gamma, beta = r(C, sc=0.1), 1 + r(C, sc=0.2), r(C, sc=0.2)
    ws = pack_block_head_stream(wi, wqkv, dt)
    wi_h, wqkv_h = wi.to(dt).contiguous(), wqkv.to(dt).contiguous()
    pe = r(24, C, sc=0.5)
    gn_g, gn_b = 1 + r(C, sc=0.2), r(C, sc=0.2)
    pool_x = [torch.randn(n, 64, 64, C, device=dev) for _ in range(4)]          # 4 x 252 MB fp32
    pool_a = [(torch.randn(M, C, device=dev).to(dt), torch.randn(M, C, device=dev)) for _ in range(4)]
    stats = ops.group_norm_stats(pool_x[0], groups=32, eps=1e-6, dtype=dt)
    ab = ops.group_norm_affine(stats, gn_g, gn_b, C, groups=32)

    def old_gn(x):
        gq, _ = ops.group_norm_apply(x, stats, gn_g, gn_b, groups=32, silu=False, dtype=dt)
        t, n1 = ops.gemm(gq.view(-1, C), wi_h, bias=bi, out_f32=True, ln=dict(gamma=gamma, beta=beta, eps=1e-5))
        return ops.gemm(n1, wqkv_h)

    def new_gn(x):
        return ops.block_head_fused(ws, bi, gamma, beta, 1e-5, x=x.view(-1, C), gn_ab=ab, rows_per_img=HW)

    ln_pe = dict(gamma=gamma, beta=beta, eps=1e-5, pe=pe, rows_per_frame=HW, pe_frames=24)

    def old_a(p):
        t, n1 = ops.gemm(p[0], wi_h, bias=bi, residual=p[1], out_f32=True, ln=ln_pe)
        return ops.gemm(n1, wqkv_h)

    def new_a(p):
        return ops.block_head_fused(ws, bi, gamma, beta, 1e-5, a=p[0], residual=p[1], pe=pe, rows_per_frame=HW, pe_frames=24)

    fl = 2 * M * C * 4 * C
    print(f"# M = {M}, C = {C}; {fl / 1e9:.1f} GFLOP per head; ms per call, best of 3 interleaved rounds, cold inputs")
    best = {}
    for _ in range(3):
        for name, fn, pool in (("gn   3 launches", old_gn, pool_x), ("gn   fused head", new_gn, pool_x),
                               ("a+res 2 launches", old_a, pool_a), ("a+res fused head", new_a, pool_a)):
            t = timed(fn, pool)
            best[name] = min(best.get(name, 1e9), t)
    for k, v in best.items():
        print(f"{k:18s} {v:7.3f} ms   {fl / v / 1e9:7.0f} TF/s")
    # the two forms agree
    y, q = new_gn(pool_x[0])
    q_old = old_gn(pool_x[0])
    print(f"# fused vs 3 launches, qkv rel-L2: {float((q.float() - q_old.float()).norm() / q_old.float().norm()):.2e}")


if __name__ == "__main__":
    main()
