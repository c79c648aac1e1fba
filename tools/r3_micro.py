"""Round-3 micro A/B on one MI355X (tune build: the legacy kernels stay selectable through the environment).
  python tools/r3_micro.py [temporal] [vae_attn]
temporal : temporal attention, first form (2-byte V gathers) vs second form (16-byte accesses + transposed LDS reads)
vae_attn : VAE mid-block attention at d = 512: flash kernel vs GEMM + row softmax + GEMM per image"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIMO_HIP_LIB", os.path.join(ROOT, "mimo_amd", "libmimo_hip_tune.so"))
from mimo_amd import ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters


def ab(fn, env_a, env_b, rounds=3):
    best = [float("inf"), float("inf")]
    for _ in range(rounds):
        for i, env in enumerate((env_a, env_b)):
            for k in set(env_a) | set(env_b):
                os.environ.pop(k, None)
            os.environ.update(env)
            best[i] = min(best[i], timeit(fn))
    for k in set(env_a) | set(env_b):
        os.environ.pop(k, None)
    return best


def temporal(dev, dt):
    print("temporal attention (b = 2, F = 24): first form vs second form; GB/s = (3 reads + 1 write) x rows x C x 2 B")
    for (HW, C) in [(4096, 320), (1024, 640), (256, 1280), (64, 1280), (9216, 320)]:
        b, F, heads = 2, 24, 8
        M = b * F * HW
        qkv = torch.randn(M, 3 * C, device=dev).to(dt)
        fn = lambda: ops.temporal_attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], b, F, HW, heads)
        old, new = ab(fn, {"MIMO_TATTN_LEGACY": "1"}, {})
        byt = 4 * M * C * 2
        print(f"  HW {HW:5d} C {C:5d}: first {old:7.3f} ms ({byt/old/1e6:7.0f} GB/s)   second {new:7.3f} ms ({byt/new/1e6:7.0f} GB/s)", flush=True)


def vae_attn(dev, dt):
    from mimo_amd.modules import Ctx
    from mimo_amd.vae import VaeMidBlock
    print("VAE mid-block attention core, d = 512, one head: flash vs unfused (per launch group of n images)")
    blk = VaeMidBlock(512, 32, 1e-6).to(dev)
    for (n, N) in [(8, 4096), (4, 9216)]:
        C = 512
        qkv = torch.randn(n, N, 3 * C, device=dev).to(dt) * 0.3
        ctx = Ctx(dt, n, 1)
        f1 = lambda: ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], 1)
        f2 = lambda: blk._attention_unfused(ctx, qkv, n, N, C)
        t1, t2 = timeit(f1, iters=5, warm=1), timeit(f2, iters=5, warm=1)
        fl = 4 * n * N * N * C
        print(f"  n {n} N {N}: flash {t1:8.3f} ms ({fl/t1/1e9:6.0f} TF/s)   unfused {t2:8.3f} ms ({fl/t2/1e9:6.0f} TF/s)", flush=True)
        a, bb = f1().float(), f2().float()
        print(f"     rel_l2(flash, unfused) = {float((a - bb).norm() / bb.norm()):.2e}")


def ff(dev, dt):
    """Level-0 feed-forward (M = 196608, C = 320): the fused kernel vs the two launches it replaces, interleaved."""
    from mimo_amd.packing import pack_ff2_kperm, pack_geglu
    print("feed-forward C = 320: mimo_ff_fused vs mimo_gemm GEGLU + mimo_gemm residual")
    C = 320
    for M in (196608, 442368):
        a = torch.randn(M, C, device=dev).to(dt)
        w1 = torch.randn(8 * C, C, device=dev) * C ** -0.5
        b1 = torch.randn(8 * C, device=dev) * 0.1
        w2 = torch.randn(C, 4 * C, device=dev) * (4 * C) ** -0.5
        b2 = torch.randn(C, device=dev) * 0.1
        res = torch.randn(M, C, device=dev)
        w1p, b1p = pack_geglu(w1, b1, dt)
        w2k, w2h = pack_ff2_kperm(w2, dt), w2.to(dt).contiguous()
        f1 = lambda: ops.ff_fused(a, w1p, b1p, w2k, b2, res)
        f2 = lambda: ops.gemm(ops.gemm(a, w1p, bias=b1p, geglu=True), w2h, bias=b2, residual=res)
        best = [1e9, 1e9]
        for _ in range(3):
            best[0] = min(best[0], timeit(f1, iters=10, warm=2))
            best[1] = min(best[1], timeit(f2, iters=10, warm=2))
        fl = 2 * M * C * 12 * C
        d = float((f1().float() - f2().float()).norm() / f2().float().norm())
        print(f"  M {M}: fused {best[0]:7.3f} ms ({fl/best[0]/1e9:6.0f} TF/s)   two launches {best[1]:7.3f} ms ({fl/best[1]/1e9:6.0f} TF/s)   rel_l2 {d:.2e}", flush=True)
        from mimo_amd.packing import pack_proj_tail
        wp = torch.randn(C, C, device=dev) * C ** -0.5
        bp = torch.randn(C, device=dev) * 0.1
        x = torch.randn(M, C, device=dev)
        wpk, wph = pack_proj_tail(wp, dt), wp.to(dt).contiguous()
        g1 = lambda: ops.ff_proj_fused(a, w1p, b1p, w2k, b2, res, wpk, bp, x)
        g2 = lambda: ops.gemm(f2(), wph, bias=bp, residual=x, out_f32=True)
        bb = [1e9, 1e9]
        for _ in range(3):
            bb[0] = min(bb[0], timeit(g1, iters=10, warm=2))
            bb[1] = min(bb[1], timeit(g2, iters=10, warm=2))
        d = float((g1() - g2()).norm() / g2().norm())
        print(f"     + proj_out: fused {bb[0]:7.3f} ms   three launches {bb[1]:7.3f} ms   rel_l2 {d:.2e}", flush=True)


def attn40_prio(dev, dt):
    """d = 40 spatial attention at the level-0 shapes (24 uncond images over N keys + 24 cond images over 2 N keys):
    s_setprio hints around the MFMA clusters, interleaved A/B on random data; results must be bit-identical."""
    print("attn40 s_setprio A/B (PRIO 0 none | 1 around Q.K^T and P.V | 2 around P.V only)")
    for (N, nb) in [(4096, 48), (9216, 48)]:
        C = 320
        qkv = torch.randn(nb, N, 3 * C, device=dev).to(dt)
        bank = torch.randn(N, 2 * C, device=dev).to(dt)
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        fl = 4 * N * C * ((nb // 2) * N + (nb // 2) * 2 * N)
        fn = lambda: ops.attention(q, k, v, 8, k2=bank[:, :C], v2=bank[:, C:], seg2_first_batch=nb // 2, q_prescaled=True)
        best = {}
        outs = {}
        for rnd in range(3):
            for prio in ("0", "1", "2"):
                os.environ["MIMO_ATTN40_PRIO"] = prio
                best[prio] = min(best.get(prio, 1e9), timeit(fn, iters=5, warm=1))
                if rnd == 0:
                    outs[prio] = fn().clone()
        os.environ.pop("MIMO_ATTN40_PRIO", None)
        same = all(torch.equal(outs["0"], outs[p_]) for p_ in ("1", "2"))
        print(f"  N {N}: " + "  ".join(f"PRIO {p_}: {best[p_]:7.3f} ms ({fl/best[p_]/1e9:6.0f} TF/s)" for p_ in ("0", "1", "2")) + f"   bit-identical: {same}", flush=True)


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    what = sys.argv[1:] or ["temporal", "vae_attn"]
    for w in what:
        {"temporal": temporal, "vae_attn": vae_attn, "attn40_prio": attn40_prio, "ff": ff}[w](dev, torch.float16)
