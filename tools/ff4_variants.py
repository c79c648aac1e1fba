"""Interleaved A/B of compile-time variants of csrc/ff_tail4.hip (ablations and schedule knobs): one small shared library per
-D setting (ff_fused.hip's entry points + that ff_tail4 object), all loaded into ONE process and timed in turn on the same
operands at the level-0 shape of configs[1] (M = 48 x 4096 rows, cold inputs).  Non-ablation variants are also compared bit for
bit with `base`.
    python tools/ff4_variants.py --build [name ...]     (here: hipcc cross-compiles; the libraries travel to the GPU box in-tree)
    python tools/ff4_variants.py [--only a,b]           (GPU box)"""
import ctypes
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, "tools", "_ab")
CSRC = os.path.join(ROOT, "mimo_amd", "csrc")
HIPCC = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]

VARIANTS = [
    ("old8", None),                       # ff_fused_kernel (8 waves x 256 registers): -DMIMO_FF_TAIL4_DEFAULT=0
    ("base", []),
    ("abl_no_dma", ["-DFF4_ABLATE=1"]),
    ("abl_no_geglu", ["-DFF4_ABLATE=2"]),
    ("abl_no_mfma", ["-DFF4_ABLATE=4"]),
    ("abl_no_frag", ["-DFF4_ABLATE=8"]),
    ("abl_no_barrier", ["-DFF4_ABLATE=16"]),
    ("abl_mfma_only", ["-DFF4_ABLATE=27"]),
    ("abl_no_loads", ["-DFF4_ABLATE=32"]),
    ("abl_no_stores", ["-DFF4_ABLATE=64"]),
    ("abl_no_io", ["-DFF4_ABLATE=96"]),
    ("exact", ["-DFF4_TRICKLE=0"]),
    ("dephase40", ["-DFF4_DEPHASE=40"]),
    ("dephase60", ["-DFF4_DEPHASE=60"]),
    ("dephase100", ["-DFF4_DEPHASE=100"]),
    ("dephase140", ["-DFF4_DEPHASE=140"]),
    ("dephase180", ["-DFF4_DEPHASE=180"]),
    ("touch", ["-DFF4_TOUCH=1"]),
    ("bias_init", ["-DFF4_BIAS_INIT=1"]),
    ("touch+bias_init", ["-DFF4_TOUCH=1", "-DFF4_BIAS_INIT=1"]),
    ("pf1", ["-DFF4_PF=1"]),
    ("no_fence", ["-DFF4_FENCE=0"]),
    ("mfma32", ["-DFF32_AS_FF4"]),        # tools/probes/ff_tail32.hip (32x32x16 MFMAs) in the place of ff_tail4.hip
    ("mfma32_nofence", ["-DFF32_AS_FF4", "-DFF32_FENCE=0"]),
    ("abl32_no_dma", ["-DFF32_AS_FF4", "-DFF32_ABLATE=1"]),
    ("abl32_no_geglu", ["-DFF32_AS_FF4", "-DFF32_ABLATE=2"]),
    ("abl32_no_mfma", ["-DFF32_AS_FF4", "-DFF32_ABLATE=4"]),
    ("abl32_no_frag", ["-DFF32_AS_FF4", "-DFF32_ABLATE=8"]),
    ("abl32_no_barrier", ["-DFF32_AS_FF4", "-DFF32_ABLATE=16"]),
    ("abl32_mfma_only", ["-DFF32_AS_FF4", "-DFF32_ABLATE=27"]),
]
EXTRA = {}   # name -> flags, filled from the command line: --def name=-DX=1,-DY=2


def lib_path(name):
    return os.path.join(VDIR, f"libff4_{name}.so")


def build(names):
    os.makedirs(VDIR, exist_ok=True)
    objs = {}
    for d in (0, 1):
        objs[d] = os.path.join(VDIR, f"ff_fused_d{d}.o")
        src = os.path.join(CSRC, "ff_fused.hip")
        if not os.path.exists(objs[d]) or os.path.getmtime(objs[d]) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(CSRC, "ff_fused.hip.h"))):
            r = subprocess.run(HIPCC + [f"-DMIMO_FF_TAIL4_DEFAULT={d}", "-DMIMO_FF_TAIL4_MODE0", "-c", src, "-o", objs[d]], capture_output=True, text=True)
            if r.returncode:
                raise RuntimeError(r.stderr[-3000:])

    def one(v):
        name, flags = v
        o = os.path.join(VDIR, f"ff_tail4_{name}.o")
        src = os.path.join(ROOT, "tools", "probes", "ff_tail32.hip") if flags and "-DFF32_AS_FF4" in flags else os.path.join(CSRC, "ff_tail4.hip")
        r = subprocess.run(HIPCC + ["-fno-slp-vectorize", "-Wno-inline-asm"] + (flags or []) + ["-c", src, "-o", o],
                           capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(r.stderr[-3000:])
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path(name), objs[0 if flags is None else 1], o],
                           capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(r.stderr[-3000:])
        os.remove(o)
        return name

    todo = [v for v in VARIANTS + list(EXTRA.items()) if not names or v[0] in names]
    with ThreadPoolExecutor(max_workers=6) as ex:
        print("built", list(ex.map(one, todo)))


def main():
    args = sys.argv[1:]
    for a in list(args):
        if a.startswith("--def"):
            k, v = args[args.index(a) + 1].split("=", 1)
            EXTRA[k] = v.split(",")
    if "--build" in args:
        names = [a for a in args[args.index("--build") + 1:] if not a.startswith("-") and "=" not in a]
        return build(names)
    import torch
    from mimo_amd import lib as L, ops
    from tools.ff4_check import C, weights
    only = args[args.index("--only") + 1].split(",") if "--only" in args else None
    present = sorted(f[7:-3] for f in os.listdir(VDIR) if f.startswith("libff4_") and f.endswith(".so"))
    order = [n for n, _ in VARIANTS if n in present] + [n for n in present if n not in dict(VARIANTS)]
    names = [n for n in order if not only or n in only]
    dev = torch.device("cuda:0")
    dt = torch.float16
    w = weights(dev, dt)
    M, HW = 48 * 4096, 4096
    pool = [(torch.randn(M, C, device=dev).to(dt), torch.randn(M, C, device=dev), torch.randn(M, C, device=dev)) for _ in range(3)]
    ib = torch.randn(48, C, device=dev)
    out_h = torch.empty(M, C, device=dev, dtype=dt)
    out_f = torch.empty(M, C, device=dev)
    cs = torch.empty(M // 32, 2, C, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    libs = {}
    for n in names:
        lb = ctypes.CDLL(lib_path(n))
        for f in ("mimo_ff_fused", "mimo_block_tail_fused"):
            getattr(lb, f).argtypes = L.SIGNATURES[f]
            getattr(lb, f).restype = ctypes.c_int
        libs[n] = lb

    def ff(lb, p):
        rc = lb.mimo_ff_fused(ops.dt_code(dt), p[0].data_ptr(), C, w["w1p"].data_ptr(), w["b1p"].data_ptr(), w["w2k"].data_ptr(), w["b2"].data_ptr(),
                              p[1].data_ptr(), C, out_h.data_ptr(), C, M, C, st)
        assert rc == 0, rc

    def tail(lb, p):
        rc = lb.mimo_block_tail_fused(ops.dt_code(dt), p[0].data_ptr(), C, w["ws"].data_ptr(), w["bo"].data_ptr(), ib.data_ptr(), C, HW,
                                      p[1].data_ptr(), C, w["gamma"].data_ptr(), w["beta"].data_ptr(), 1e-5, w["b1p"].data_ptr(),
                                      w["w2k"].data_ptr(), w["b2"].data_ptr(), w["bp"].data_ptr(), p[2].data_ptr(), C, out_f.data_ptr(), C, M, C,
                                      cs.data_ptr(), st)
        assert rc == 0, rc

    def timed(fn, lb, iters=10):
        for i in range(2):
            fn(lb, pool[i % 3])
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for i in range(iters):
            fn(lb, pool[i % 3])
        ev[1].record()
        torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) / iters

    best = {}
    for _ in range(3):
        for n in names:
            for tag, fn in (("ff", ff), ("tail", tail)):
                best[(n, tag)] = min(best.get((n, tag), 1e9), timed(fn, libs[n]))
    ref, ref8 = {}, {}
    if "base" in libs:
        ff(libs["base"], pool[0]); ref["ff"] = out_h.clone()
        tail(libs["base"], pool[0]); ref["tail"] = out_f.clone()
    if "old8" in libs:
        ff(libs["old8"], pool[0]); ref8["ff"] = out_h.clone()
        tail(libs["old8"], pool[0]); ref8["tail"] = out_f.clone()
    print(f"# M = {M}, C = {C}, fp16; ms per launch (best of 3 interleaved rounds, cold inputs): mimo_ff_fused | mimo_block_tail_fused")
    for n in names:
        eq = ""
        if ref and not n.startswith("abl"):
            out_h.zero_(); ff(libs[n], pool[0]); e1 = torch.equal(out_h, ref["ff"])
            out_f.zero_(); tail(libs[n], pool[0]); e2 = torch.equal(out_f, ref["tail"])
            d = float((out_f - ref["tail"]).norm() / ref["tail"].norm())
            eq = f"   bit-equal to base: {e1 and e2}" + ("" if e1 and e2 else f"  (tail rel-L2 vs base {d:.2e})")
            if ref8:
                eq += f"   bit-equal to old8 (ff_fused_kernel): {torch.equal(out_h, ref8['ff']) and torch.equal(out_f, ref8['tail'])}"
        print(f"{n:18s} {best[(n, 'ff')]:7.3f}  {best[(n, 'tail')]:7.3f}{eq}", flush=True)


if __name__ == "__main__":
    main()
