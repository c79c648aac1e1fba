"""Per-kernel micro-benchmarks at the config-2 shapes of the denoising UNet.
Usage (GPU box): python tools/microbench.py [--dtype fp16|bf16]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the tuning knobs exist only in the -DMIMO_TUNE build of the library (python -m mimo_amd.build --tune)
if "--ab" in sys.argv or "--attn-ab" in sys.argv or os.environ.get("MIMO_USE_TUNE_LIB"):
    os.environ.setdefault("MIMO_HIP_LIB", os.path.join(ROOT, "mimo_amd", "libmimo_hip_tune.so"))
from mimo_amd import lib as L, ops  # noqa: E402
from mimo_amd.packing import pack_conv, pack_geglu  # noqa: E402


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters * 1e-3


def ab_table(settings, dt, dev, n):
    """Interleaved A/B: every shape is timed under every knob setting in turn, 3 rounds, best median kept."""
    knobs = ("MIMO_GEMM_CFG", "MIMO_GEMM_STAGGER", "MIMO_CONV_TAP_INNER", "MIMO_GEMM_SPLITK", "MIMO_GEMM_ABLATE", "MIMO_GEMM_BM", "MIMO_GEMM_PERSIST", "MIMO_GEMM_STREAM", "MIMO_STREAM_ABLATE")
    cases = []
    for (hw, cin, cout) in [(64, 320, 320), (32, 640, 640), (16, 1280, 1280), (8, 1280, 1280), (8, 2560, 1280), (64, 960, 320), (16, 2560, 1280)]:
        x = torch.randn(n, hw, hw, cin, device=dev).to(dt)
        w = pack_conv(torch.randn(cout, cin, 3, 3, device=dev) * 0.02, dt)
        b = torch.zeros(cout, device=dev)
        cases.append((f"conv3x3 {hw}x{hw} {cin}->{cout}", (lambda x=x, w=w, b=b, cout=cout: ops.conv2d(x, w, cout, bias=b, out_f32=True)),
                      2 * n * hw * hw * cout * 9 * cin))
    for (nn, hw, cin, cout) in [(8, 512, 128, 128), (8, 256, 256, 256), (8, 128, 512, 512)]:  # VAE decoder convs
        x = torch.randn(nn, hw, hw, cin, device=dev).to(dt)
        w = pack_conv(torch.randn(cout, cin, 3, 3, device=dev) * 0.02, dt)
        b = torch.zeros(cout, device=dev)
        cases.append((f"vae conv3x3 n{nn} {hw}x{hw} {cin}->{cout}", (lambda x=x, w=w, b=b, cout=cout: ops.conv2d(x, w, cout, bias=b, out_f32=True)),
                      2 * nn * hw * hw * cout * 9 * cin))
    for (M, N, K) in [(196608, 320, 320), (196608, 960, 320), (196608, 320, 1280), (49152, 640, 640), (49152, 640, 2560),
                      (49152, 1920, 640), (12288, 3840, 1280),
                      (12288, 1280, 1280), (12288, 1280, 5120), (3072, 1280, 1280), (3072, 1280, 5120)]:
        A = torch.randn(M, K, device=dev).to(dt)
        W = (torch.randn(N, K, device=dev) * 0.02).to(dt)
        R = torch.randn(M, N, device=dev)
        cases.append((f"gemm M{M} N{N} K{K}", (lambda A=A, W=W: ops.gemm(A, W)), 2 * M * N * K))
        if N in (320, 640, 1280) and M >= 12288:
            cases.append((f"gemm+res32 M{M} N{N} K{K}", (lambda A=A, W=W, R=R: ops.gemm(A, W, residual=R, out_f32=True)), 2 * M * N * K))
    for (M, dim) in [(196608, 320), (49152, 640), (12288, 1280)]:
        A = torch.randn(M, dim, device=dev).to(dt)
        wp, bp = pack_geglu(torch.randn(8 * dim, dim, device=dev) * 0.02, torch.zeros(8 * dim, device=dev), dt)
        cases.append((f"geglu-ff1 M{M} dim{dim}", (lambda A=A, wp=wp, bp=bp: ops.gemm(A, wp, bias=bp, geglu=True)), 2 * M * 8 * dim * dim))
    print("setting index: " + "  ".join(f"[{i}] {s or 'default'}" for i, s in enumerate(settings)))
    print(f"{'case':36s} " + " ".join(f"{'['+str(i)+'] ms':>9s} {'TF/s':>7s}" for i in range(len(settings))))
    for name, fn, fl in cases:
        best = [float("inf")] * len(settings)
        for _ in range(3):
            for i, sset in enumerate(settings):
                for k in knobs:
                    os.environ.pop(k, None)
                for kv in filter(None, sset.split(",")):
                    k, v = kv.split("=")
                    os.environ[k] = v
                L.call("mimo_reload_tuning")
                best[i] = min(best[i], timeit(fn, iters=20, warm=2))
        print(f"{name:36s} " + " ".join(f"{t*1e3:9.3f} {fl/t/1e12:7.1f}" for t in best), flush=True)
    for k in knobs:
        os.environ.pop(k, None)
    L.call("mimo_reload_tuning")


def med_interleaved(fns, rounds=5, iters=10):
    """Interleaved timing of several variants of one case: `rounds` rounds, each variant timed in turn; min and median."""
    ts = [[] for _ in fns]
    for _ in range(rounds):
        for i, fn in enumerate(fns):
            ts[i].append(timeit(fn, iters=iters, warm=2))
    return [(min(t), sorted(t)[len(t) // 2]) for t in ts]


def attn_ab(dt, dev):
    """Spatial attention d = 40: legacy kernel vs attn40_kernel program orders (tune build: MIMO_ATTN40_* knobs)."""
    variants = [("legacy", {"MIMO_ATTN40_LEGACY": "1"}), ("v3 4 waves", {"MIMO_ATTN40_NW": "4"}), ("v3 8 waves", {"MIMO_ATTN40_NW": "8"}),
                ("v3 4w no-mem", {"MIMO_ATTN40_NW": "4", "MIMO_ATTN40_ABLATE": "1"}),
                ("v3 4w no-wait", {"MIMO_ATTN40_NW": "4", "MIMO_ATTN40_ABLATE": "2"}),
                ("v3 4w regstage", {"MIMO_ATTN40_NW": "4", "MIMO_ATTN40_STAGE": "1"})]
    for (N, C, nb) in [(4096, 320, 48), (1024, 320, 48), (9604, 320, 8)]:
        qkv = torch.randn(nb, N, 3 * C, device=dev).to(dt)
        bank = torch.randn(N, 2 * C, device=dev).to(dt)
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        fl = 4 * N * C * ((nb // 2) * N + (nb // 2) * 2 * N)

        def mk(env):
            def f():
                for kk in ("MIMO_ATTN40_LEGACY", "MIMO_ATTN40_NW", "MIMO_ATTN40_ABLATE", "MIMO_ATTN40_STAGE"):
                    os.environ.pop(kk, None)
                os.environ.update(env)
                return ops.attention(q, k, v, 8, k2=bank[:, :C], v2=bank[:, C:], seg2_first_batch=nb // 2, q_prescaled=True)
            return f
        fns = [mk(env) for _, env in variants]
        outs = [f().float() for f in fns]
        # the same work with K / V tiles CONTIGUOUS in memory (head-major [B*heads, N, 40]: a tile DMA is 8 full 128-byte
        # lines instead of ~20 partial ones of 80 bytes at the 1920-byte token pitch): does the L1 request count matter?
        qh = torch.randn(nb * 8, N, 40, device=dev).to(dt)
        kh, vh = torch.randn(nb * 8, N, 40, device=dev).to(dt), torch.randn(nb * 8, N, 40, device=dev).to(dt)
        bh = torch.randn(N, 80, device=dev).to(dt)

        def contiguous_kv():
            for kk in ("MIMO_ATTN40_LEGACY", "MIMO_ATTN40_NW", "MIMO_ATTN40_ABLATE", "MIMO_ATTN40_STAGE"):
                os.environ.pop(kk, None)
            return ops.attention(qh, kh, vh, 1, k2=bh[:, :40], v2=bh[:, 40:], seg2_first_batch=nb * 4, q_prescaled=True)
        fns.append(contiguous_kv)
        names = [n for n, _ in variants] + ["v3 head-major"]
        outs.append(outs[0])
        contiguous_kv()
        res = med_interleaved(fns)
        for name, (tmin, tmed), o in zip(names, res, outs):
            err = float((o - outs[0]).norm() / outs[0].norm())
            print(f"attn N{N} d40 b{nb} {name:13s}: min {tmin*1e3:7.3f} ms med {tmed*1e3:7.3f} ms  {fl/tmin/1e12:7.1f} TF/s  rel-L2 vs legacy {err:.1e}", flush=True)
    for kk in ("MIMO_ATTN40_LEGACY", "MIMO_ATTN40_NW", "MIMO_ATTN40_ABLATE", "MIMO_ATTN40_STAGE"):
        os.environ.pop(kk, None)


def fused_paths(dt, dev):
    """Round-2 fusions against the launches they replace: epilogue column statistics vs the GroupNorm statistics pass,
    LayerNorm in the GEMM epilogue vs GEMM + mimo_layer_norm."""
    n = 48
    for (hw, C) in [(64, 320), (32, 640), (16, 1280)]:
        x = torch.randn(n, hw, hw, C, device=dev).to(dt)
        w = pack_conv(torch.randn(C, C, 3, 3, device=dev) * 0.02, dt)
        b = torch.zeros(C, device=dev)
        g = torch.ones(C, device=dev)
        plain = lambda: ops.conv2d(x, w, C, bias=b, out_f32=True)
        stats = lambda: ops.conv2d(x, w, C, bias=b, out_f32=True, colstats=True)
        y = stats()
        y0 = plain()
        gn_cols = lambda: ops.group_norm(y, g, b, silu=True, dtype=dt)
        gn_pass = lambda: ops.group_norm(y0, g, b, silu=True, dtype=dt)
        r = med_interleaved([plain, stats, gn_pass, gn_cols])
        print(f"conv3x3 {hw}x{hw} C{C}: plain {r[0][0]*1e3:.3f} ms | +colstats {r[1][0]*1e3:.3f} ms || GN(stats pass + apply) {r[2][0]*1e3:.3f} ms | "
              f"GN(cols + apply) {r[3][0]*1e3:.3f} ms", flush=True)
    M, C = 196608, 320
    A = torch.randn(M, C, device=dev).to(dt)
    W = (torch.randn(C, C, device=dev) * 0.05).to(dt)
    R = torch.randn(M, C, device=dev)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    pe = torch.randn(32, C, device=dev)
    sep = lambda: ops.layer_norm(ops.gemm(A, W, bias=b, residual=R, out_f32=True), g, b, dtype=dt)
    fus = lambda: ops.gemm(A, W, bias=b, residual=R, out_f32=True, ln=dict(gamma=g, beta=b))
    fus_pe = lambda: ops.gemm(A, W, bias=b, residual=R, out_f32=True, ln=dict(gamma=g, beta=b, pe=pe, rows_per_frame=4096, pe_frames=24))
    only = lambda: ops.gemm(A, W, bias=b, residual=R, out_f32=True)
    r = med_interleaved([only, sep, fus, fus_pe])
    print(f"gemm+res32 M{M} N320 K320: gemm {r[0][0]*1e3:.3f} ms | gemm + layer_norm {r[1][0]*1e3:.3f} ms | fused LN {r[2][0]*1e3:.3f} ms | "
          f"fused LN+pe {r[3][0]*1e3:.3f} ms", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--norms", action="store_true", help="only the norm kernels")
    ap.add_argument("--attn", action="store_true", help="only the spatial attention kernel")
    ap.add_argument("--mm", action="store_true", help="only the gemm_kernel family (convs, linears, GEGLU)")
    ap.add_argument("--ab", default="", help="A/B table of the gemm_kernel family over tuning-knob settings, e.g. "
                    "'MIMO_GEMM_CFG=3;MIMO_GEMM_CFG=4,MIMO_GEMM_STAGGER=1' (settings separated by ';')")
    ap.add_argument("--vae", action="store_true", help="time VAE encode/decode + pose guider on 8 frames at 512x512 instead")
    ap.add_argument("--attn-ab", action="store_true", help="d = 40 attention: legacy vs attn40_kernel program orders (tune build)")
    ap.add_argument("--fused", action="store_true", help="epilogue column statistics / fused LayerNorm vs the launches they replace")
    a = ap.parse_args()
    dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    dev = torch.device("cuda:0")
    n = 48
    print(f"# dtype={a.dtype} device={torch.cuda.get_device_name(0)}")
    if a.ab:
        return ab_table(a.ab.split(";"), dt, dev, n)
    if a.attn_ab:
        return attn_ab(dt, dev)
    if a.fused:
        return fused_paths(dt, dev)
    if a.vae:
        from mimo_amd.vae import AutoencoderKL, PoseGuider
        with torch.device(dev):
            vae, pg = AutoencoderKL(), PoseGuider()
        torch.nn.init.normal_(pg.conv_out.weight, std=0.02)
        for m in (vae, pg):
            m.to(dtype=dt)
            m.compute_dtype = dt
        img = torch.rand(8, 512, 512, 8, device=dev).to(dt)
        img[..., 3:] = 0
        z = torch.randn(8, 64, 64, 8, device=dev).to(dt)
        z[..., 4:] = 0
        for max_hw in (ops.COLSTATS_MAX_HW, 1 << 30):  # second pass: GroupNorm column statistics from the conv epilogues at every size
            ops.COLSTATS_MAX_HW = max_hw
            print(f"# COLSTATS_MAX_HW = {max_hw}")
            for name, fn, fl in (("vae.encode 8x512^2", lambda: vae.encode_tokens(img), 8 * 1.1167e12),
                                 ("vae.decode 8x64^2->512^2", lambda: vae.decode_tokens(z), 8 * 2.5145e12),
                                 ("pose_guider 8x512^2", lambda: pg.run_tokens(img), 8 * 0.0147e12)):
                t = timeit(fn, iters=3, warm=1)
                print(f"{name}: {t*1e3:8.2f} ms  {fl/t/1e12:7.1f} TF/s (algorithmic)", flush=True)
        return
    # --- 3x3 convs (n, hw, cin, cout)
    for (hw, cin, cout) in [] if (a.norms or a.attn) else [(64, 320, 320), (32, 640, 640), (16, 1280, 1280), (8, 1280, 1280), (64, 960, 320), (16, 2560, 1280)]:
        x = torch.randn(n, hw, hw, cin, device=dev).to(dt)
        w = pack_conv(torch.randn(cout, cin, 3, 3, device=dev) * 0.02, dt)
        b = torch.zeros(cout, device=dev)
        t = timeit(lambda: ops.conv2d(x, w, cout, bias=b, out_f32=True))
        fl = 2 * n * hw * hw * cout * 9 * cin
        print(f"conv3x3 n{n} {hw}x{hw} {cin}->{cout}: {t*1e3:8.3f} ms  {fl/t/1e12:7.1f} TF/s")
    # --- GEMMs (M, N, K)
    for (M, N, K) in [] if (a.norms or a.attn) else [(196608, 320, 320), (196608, 960, 320), (196608, 320, 1280), (49152, 640, 640), (12288, 1280, 1280), (12288, 1280, 5120), (3072, 1280, 1280)]:
        A = torch.randn(M, K, device=dev).to(dt)
        W = (torch.randn(N, K, device=dev) * 0.02).to(dt)
        t = timeit(lambda: ops.gemm(A, W))
        print(f"gemm M{M} N{N} K{K}: {t*1e3:8.3f} ms  {2*M*N*K/t/1e12:7.1f} TF/s")
    for (M, dim) in [] if (a.norms or a.attn) else [(196608, 320), (49152, 640), (12288, 1280)]:
        A = torch.randn(M, dim, device=dev).to(dt)
        wp, bp = pack_geglu(torch.randn(8 * dim, dim, device=dev) * 0.02, torch.zeros(8 * dim, device=dev), dt)
        t = timeit(lambda: ops.gemm(A, wp, bias=bp, geglu=True))
        print(f"geglu-ff1 M{M} dim{dim}: {t*1e3:8.3f} ms  {2*M*8*dim*dim/t/1e12:7.1f} TF/s")
    if a.mm:
        return
    # --- spatial attention, 24 uncond + 24 cond (bank)
    for (N, C) in [] if a.norms else [(4096, 320), (1024, 640), (256, 1280), (64, 1280)]:
        qkv = torch.randn(n, N, 3 * C, device=dev).to(dt)
        bank = torch.randn(N, 2 * C, device=dev).to(dt)
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        t = timeit(lambda: ops.attention(q, k, v, 8, k2=bank[:, :C], v2=bank[:, C:], seg2_first_batch=24))
        fl = 4 * N * C * (24 * N + 24 * 2 * N)
        print(f"attn N{N} C{C} d{C//8}: {t*1e3:8.3f} ms  {fl/t/1e12:7.1f} TF/s")
        t = timeit(lambda: ops.attention(q, k, v, 8, k2=bank[:, :C], v2=bank[:, C:], seg2_first_batch=24, q_prescaled=True))
        print(f"attn N{N} C{C} d{C//8} (q prescaled): {t*1e3:8.3f} ms  {fl/t/1e12:7.1f} TF/s")
    if a.attn:
        return
    # --- temporal attention
    for (HW, C) in [(4096, 320), (1024, 640), (256, 1280), (64, 1280)]:
        qkv = torch.randn(2 * 24 * HW, 3 * C, device=dev).to(dt)
        t = timeit(lambda: ops.temporal_attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], 2, 24, HW, 8))
        by = qkv.numel() * 2 + qkv.shape[0] * C * 2
        print(f"temporal HW{HW} C{C}: {t*1e3:8.3f} ms  {by/t/1e9:7.1f} GB/s")
    # --- norms
    for (hw, C) in [(64, 320), (32, 640), (16, 1280)]:
        x = torch.randn(n, hw, hw, C, device=dev)
        g = torch.ones(C, device=dev)
        b = torch.zeros(C, device=dev)
        t = timeit(lambda: ops.group_norm(x, g, b, silu=True, dtype=dt))
        print(f"groupnorm+silu n{n} {hw}x{hw} C{C}: {t*1e3:8.3f} ms  {x.numel()*(4+4+2)/t/1e9:7.1f} GB/s(2R+1W)")
        st = torch.empty(n, 32, 2, device=dev)
        o = torch.empty(x.shape, device=dev, dtype=dt)
        t = timeit(lambda: L.call("mimo_group_norm_apply", x.data_ptr(), C, None, 0, 1, ops.dt_code(dt), n, hw * hw, 32,
                                  st.data_ptr(), g.data_ptr(), b.data_ptr(), 1, o.data_ptr(), None, None))
        print(f"  gn apply only: {t*1e3:8.3f} ms  {x.numel()*6/t/1e9:7.1f} GB/s(1R+1W)")
        t = timeit(lambda: ops.layer_norm(x.view(-1, C), g, b, dtype=dt))
        print(f"layernorm rows{n*hw*hw} C{C}: {t*1e3:8.3f} ms  {x.numel()*6/t/1e9:7.1f} GB/s")


if __name__ == "__main__":
    main()
