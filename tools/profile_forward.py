"""One-process driver for rocprofv3: builds the full-size denoising UNet (+ reference UNet for the banks) and runs
a few denoising forwards on a CFG batch of 2 x 24 latent frames (config 2).  Usage on the GPU box:
  cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OUT -- python tools/profile_forward.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    dtype = torch.bfloat16 if "--bf16" in sys.argv else torch.float16
    dev = torch.device("cuda:0")
    pipe = bench.build_pipeline(dev, dtype)
    t, fl, n, fam = bench.measure_forward(pipe, dev, dtype, 512, iters=3)
    print(f"forward {t*1e3:.1f} ms  {fl/1e12:.2f} TFLOP  {fl/t/1e12:.1f} TF/s  {n} launches", flush=True)
    for k, d in fam.items():
        print(f"  {k}: {d['launches']} launches, {d['ms']:.2f} ms, {d['flops']/1e12:.2f} TFLOP, {d['flops']/d['ms']/1e9:.1f} TF/s, "
              f"avg {d['ms']*1e3/d['launches']:.1f} us", flush=True)
    torch.cuda.synchronize()
    # under rocprofv3 the interpreter can hang after the tool has written its output: give teardown 60 s, then leave
    import threading
    t = threading.Timer(60.0, os._exit, [0])
    t.daemon = True
    t.start()


if __name__ == "__main__":
    main()
