"""Thin convolutions, old path (implicit-GEMM kernel) vs new (GPU box, tune library for the environment knob):
  thin input  (pose guider, VAE conv_in): mimo_conv2d with MIMO_THIN_CONV=0 | 1 (csrc/thinconv.hip)
  thin output (conv_out of the UNet / VAE decoder): ops.conv2d vs ops.conv3x3_thin_out (GEMM + mimo_conv3x3_tapsum)
`bytes` = input + output + weights once; TB/s = bytes / time; x byte time = time / (bytes at 4.4 TB/s).
    python tools/thin_bench.py > profiles/r4_thin_conv_bench.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MIMO_HIP_LIB", os.path.join(ROOT, "mimo_amd", "libmimo_hip_tune.so"))
import torch  # noqa: E402

sys.path.insert(0, ROOT)
from mimo_amd import ops  # noqa: E402
from mimo_amd.packing import pack_conv, pack_conv_taps  # noqa: E402


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        st.record()
        for _ in range(iters):
            fn()
        en.record()
        torch.cuda.synchronize()
        best = min(best, st.elapsed_time(en) / iters * 1e-3)
    return best


def main():
    dev, dt = torch.device("cuda:0"), torch.float16
    print(f"{'layer':44s} {'old ms':>8s} {'new ms':>8s} {'new/old':>8s} {'TB/s new':>9s} {'x byte time':>11s}")
    thin_in = [("pose conv_in 8->16 512^2 n24", 24, 512, 512, 8, 16, 1, True), ("pose 16->16 512^2 n24", 24, 512, 512, 16, 16, 1, True),
               ("pose 16->32 s2 512^2 n24", 24, 512, 512, 16, 32, 2, True), ("pose 32->32 256^2 n24", 24, 256, 256, 32, 32, 1, True),
               ("pose 32->96 s2 256^2 n24", 24, 256, 256, 32, 96, 2, True), ("vae conv_in 8->128 512^2 n1", 1, 512, 512, 8, 128, 1, False)]
    for label, n, H, W, cin, cout, stride, silu in thin_in:
        x = torch.randn(n, H, W, cin, device=dev).to(dt)
        w = pack_conv(torch.randn(cout, cin, 3, 3, device=dev) * 0.1, dt)
        b = torch.zeros(cout, device=dev)
        run = lambda: ops.conv2d(x, w, cout, stride=stride, bias=b, silu=silu, out_f32=not silu)
        os.environ["MIMO_THIN_CONV"] = "0"
        o_old, t_old = run(), timeit(run)
        os.environ["MIMO_THIN_CONV"] = "1"
        o_new, t_new = run(), timeit(run)
        err = float((o_old.float() - o_new.float()).norm() / o_old.float().norm())
        nbytes = x.numel() * 2 + o_new.numel() * o_new.element_size() + w.numel() * 2
        print(f"{label:44s} {t_old*1e3:8.3f} {t_new*1e3:8.3f} {t_new/t_old:8.3f} {nbytes/t_new/1e12:9.2f} {t_new/(nbytes/4.4e12):11.2f}   rel diff {err:.1e}", flush=True)
    for label, n, H, W, cin, cout in [("unet conv_out 320->4 64^2 n48", 48, 64, 64, 320, 4), ("vae conv_out 128->4 512^2 n8", 8, 512, 512, 128, 4)]:
        x = torch.randn(n, H, W, cin, device=dev).to(dt)
        wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
        w, wtaps = pack_conv(wt, dt), pack_conv_taps(wt, dt)
        b = torch.zeros(cout, device=dev)
        old = lambda: ops.conv2d(x, w, cout, bias=b, out_f32=True)
        new = lambda: ops.conv3x3_thin_out(x, wtaps, cout, bias=b)
        o_old, t_old = old(), timeit(old)
        o_new, t_new = new(), timeit(new)
        err = float((o_old - o_new).norm() / o_old.norm())
        t_gemm = timeit(lambda: ops.conv2d(x, wtaps, 9 * cout, ksize=1, out_f32=True))
        os.environ["MIMO_THIN_CONV"] = "0"
        t_gemm_old = timeit(lambda: ops.conv2d(x, wtaps, 9 * cout, ksize=1, out_f32=True))
        os.environ["MIMO_THIN_CONV"] = "1"
        label += f" [tap GEMM {t_gemm*1e3:.3f} (tiled kernel {t_gemm_old*1e3:.3f})]"
        nbytes = x.numel() * 2 + o_new.numel() * 4 + w.numel() * 2
        print(f"{label:76s} {t_old*1e3:8.3f} {t_new*1e3:8.3f} {t_new/t_old:8.3f} {nbytes/t_new/1e12:9.2f} {t_new/(nbytes/4.4e12):11.2f}   rel diff {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
