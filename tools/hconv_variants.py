"""Interleaved A/B of compile-time variants of csrc/hconv.hip (one small shared library per -D setting, all loaded into
ONE process and timed in turn on the same operands: cdna_hip_programming.md rules 19 / 24).

    python tools/hconv_variants.py --build            (here: hipcc cross-compiles; the libraries travel to the GPU box in-tree)
    python tools/hconv_variants.py [--cases unet|vae|all] [--dtype fp16|bf16]      (GPU box)
"""
import argparse
import ctypes
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "tools", "_ab")
SRC = os.path.join(ROOT, "mimo_amd", "csrc", "hconv.hip")

VARIANTS = [
    ("base", []),
    ("setprio", ["-DHCONV_SETPRIO=1"]),
    ("no_fence", ["-DHCONV_FENCE=0"]),
    ("setprio+no_fence", ["-DHCONV_SETPRIO=1", "-DHCONV_FENCE=0"]),
    ("pro", ["-DHCONV_PRO=1"]),
    ("pro+setprio+nofence", ["-DHCONV_PRO=1", "-DHCONV_SETPRIO=1", "-DHCONV_FENCE=0"]),
    ("abl_no_valu", ["-DHCONV_ABLATE=1"]),
    ("abl_no_wdma", ["-DHCONV_ABLATE=4"]),
    ("abl_no_pdma", ["-DHCONV_ABLATE=8"]),
    ("abl_no_dma", ["-DHCONV_ABLATE=12"]),
    ("abl_no_valu_pdma", ["-DHCONV_ABLATE=9"]),
]


def lib_path(name):
    return os.path.join(VDIR, f"libhconv_{name.replace('+', '_')}.so")


def build():
    os.makedirs(VDIR, exist_ok=True)

    def one(v):
        name, flags = v
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.join(ROOT, "mimo_amd", "csrc")] + flags + [SRC, "-o", lib_path(name)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(r.stderr[-3000:])
        return name

    with ThreadPoolExecutor(max_workers=5) as ex:
        print("built", list(ex.map(one, VARIANTS)))


def timeit(fn, iters=10, warm=2):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--cases", default="all")
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    if a.build:
        return build()
    from mimo_amd import lib as L, ops
    from mimo_amd.packing import pack_conv
    dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    dev = torch.device("cuda:0")
    names = [n for n, _ in VARIANTS if os.path.exists(lib_path(n)) and (not a.only or n in a.only.split(","))]
    libs = []
    for n in names:
        lb = ctypes.CDLL(lib_path(n))
        lb.mimo_conv3x3_fused.argtypes = L.SIGNATURES["mimo_conv3x3_fused"]
        lb.mimo_conv3x3_fused.restype = ctypes.c_int
        libs.append(lb)
    unet = [("unet L0 320->320", 48, 64, 64, 320, 0, 320, True, False, False),
            ("unet L0 320+640->320 +raw", 48, 64, 64, 320, 640, 320, True, False, True),
            ("unet up 640->640", 48, 64, 64, 640, 0, 640, False, True, False)]
    vae = [("vae 512^2 128->128", 8, 512, 512, 128, 0, 128, True, False, False),
           ("vae 256^2 256->256", 8, 256, 256, 256, 0, 256, True, False, False),
           ("vae 128^2 512->512", 8, 128, 128, 512, 0, 512, True, False, False)]
    cases = unet if a.cases == "unet" else vae if a.cases == "vae" else unet + vae
    print(f"{'case':30s} " + " ".join(f"{n[:16]:>16s}" for n in names) + "   (ms min of 5 rounds; TF/s of base)")
    for (label, n, H, W, C1, C2, cout, gn, ups, want_raw) in cases:
        C = C1 + C2
        Hs, Ws = (H // 2, W // 2) if ups else (H, W)
        x1 = torch.randn(n, Hs, Ws, C1, device=dev)
        x2 = torch.randn(n, Hs, Ws, C2, device=dev) if C2 else None
        w = pack_conv(torch.randn(cout, C, 3, 3, device=dev) * 0.02, dt)
        b = torch.zeros(cout, device=dev)
        ab = None
        if gn:
            stats = ops.group_norm_stats(x1, groups=32, x2=x2, dtype=dt)
            ab = ops.group_norm_affine(stats, torch.ones(C, device=dev), torch.zeros(C, device=dev), C)
        out = torch.empty((n, H, W, cout), device=dev)
        raw = torch.empty((n, H, W, C), device=dev, dtype=dt) if want_raw else None
        p = L.HconvParams(n, H, W, cout, int(ups), 1, 0)
        st = torch.cuda.current_stream().cuda_stream

        def mk(lb):
            def f():
                rc = lb.mimo_conv3x3_fused(ops.dt_code(dt), x1.data_ptr(), C1, None if x2 is None else x2.data_ptr(), C2,
                                           None if ab is None else ab.data_ptr(), int(ab is not None), w.data_ptr(), w.shape[1],
                                           out.data_ptr(), ctypes.byref(p), b.data_ptr(), None, None,
                                           None if raw is None else raw.data_ptr(), None, 1.0, L.EPI_OUT_F32, st)
                assert rc == 0, rc
            return f
        fns = [mk(lb) for lb in libs]
        best = [float("inf")] * len(fns)
        for _ in range(5):
            for i, f in enumerate(fns):
                best[i] = min(best[i], timeit(f))
        fns[0]()
        ref_out = out.clone()
        bad = []
        for nm, f in zip(names, fns):
            if not nm.startswith("abl"):
                out.zero_()
                f()
                if not torch.equal(out, ref_out):
                    bad.append(nm)
        fl = 2 * n * H * W * cout * 9 * C
        if bad:
            print("   DIFFERENT RESULT:", bad)
        print(f"{label:30s} " + " ".join(f"{t*1e3:16.3f}" for t in best) + f"   {fl/best[0]/1e12:7.1f}", flush=True)


if __name__ == "__main__":
    main()
