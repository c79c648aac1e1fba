"""Per-kernel micro-benchmarks at the config-2 shapes of the denoising UNet.
Usage (GPU box): python tools/microbench.py [--dtype fp16|bf16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    knobs = ("MIMO_GEMM_CFG", "MIMO_GEMM_STAGGER", "MIMO_CONV_TAP_INNER", "MIMO_GEMM_SPLITK", "MIMO_GEMM_ABLATE", "MIMO_GEMM_BM", "MIMO_GEMM_PERSIST")
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
    a = ap.parse_args()
    dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    dev = torch.device("cuda:0")
    n = 48
    print(f"# dtype={a.dtype} device={torch.cuda.get_device_name(0)}")
    if a.ab:
        return ab_table(a.ab.split(";"), dt, dev, n)
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
        for name, fn, fl in (("vae.encode 8x512^2", lambda: vae.encode_tokens(img), 8 * 1.1167e12),
                             ("vae.decode 8x64^2->512^2", lambda: vae.decode_tokens(z), 8 * 2.5145e12),
                             ("pose_guider 8x512^2", lambda: pg.run_tokens(img), 8 * 0.0147e12)):
            t = timeit(fn, iters=3, warm=1)
            print(f"{name}: {t*1e3:8.2f} ms  {fl/t/1e12:7.1f} TF/s (algorithmic)")
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
