"""LayerNorm launch-geometry probe (tune build): fp32 rows -> half, the shapes of the denoising forward, over caps of the
persistent grid (MIMO_LN_BLOCKS blocks of 4 waves).   python tools/ln_micro.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIMO_HIP_LIB", os.path.join(ROOT, "mimo_amd", "libmimo_hip_tune.so"))
import torch  # noqa: E402

from mimo_amd import ops  # noqa: E402


def timed(fn, iters=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    caps = [int(c) for c in os.environ.get("LN_CAPS", "512,1024,1536,2048,3072,4096,8192").split(",")]
    print("rows x C          " + "".join(f"{c:>9d}" for c in caps) + "   (us per launch; ideal at 4.4 TB/s)")
    for rows, C in [(196608, 320), (49152, 640), (12288, 1280), (3072, 1280)]:
        xs = [torch.randn(rows, C, device=dev) for _ in range(max(2, int(600e6 // (rows * C * 4))))]   # rotate: no cache reuse
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        i = [0]

        def fn():
            i[0] += 1
            return ops.layer_norm(xs[i[0] % len(xs)], g, b, eps=1e-5, dtype=torch.float16)
        row = []
        for c in caps:
            os.environ["MIMO_LN_BLOCKS"] = str(c)
            row.append(timed(fn))
        print(f"{rows:7d} x {C:5d}   " + "".join(f"{t:9.1f}" for t in row) + f"   {rows*C*6/4.4e12*1e6:7.1f}", flush=True)


if __name__ == "__main__" and "--gn" not in sys.argv:
    main()


def gn_probe():
    """GroupNorm apply (+SiLU), fp32 -> half, over caps of the grid-stride grid (MIMO_STREAM_BLOCKS)."""
    dev = torch.device("cuda:0")
    caps = [int(c) for c in os.environ.get("GN_CAPS", "512,1024,2048,4096,8192,16384").split(",")]
    print("gn_apply n x hw x C   " + "".join(f"{c:>9d}" for c in caps) + "   (us per launch; ideal at 4.4 TB/s)")
    for n, hw, C in [(48, 64, 320), (48, 32, 640), (48, 16, 1280), (48, 8, 1280), (48, 64, 960), (48, 16, 2560), (8, 512, 128)]:
        xs = [torch.randn(n, hw, hw, C, device=dev) for _ in range(max(2, int(600e6 // (n * hw * hw * C * 4))))]
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        st = ops.group_norm_stats(xs[0], groups=32, eps=1e-5, dtype=torch.float16)
        i = [0]

        def fn():
            i[0] += 1
            return ops.group_norm_apply(xs[i[0] % len(xs)], st, g, b, groups=32, silu=True, dtype=torch.float16)
        row = []
        for c in caps:
            os.environ["MIMO_STREAM_BLOCKS"] = str(c)
            row.append(timed(fn))
        print(f"{n:3d} x {hw:3d}^2 x {C:5d}     " + "".join(f"{t:9.1f}" for t in row) + f"   {n*hw*hw*C*6/4.4e12*1e6:7.1f}", flush=True)


if __name__ == "__main__" and "--gn" in sys.argv:
    gn_probe()
