"""One-process driver for rocprofv3: VAE decode (and encode) of 8 frames at 512x512, three rounds.
  cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OUT -- python tools/vae_only.py"""
import os
import sys
import threading

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimo_amd.vae import AutoencoderKL  # noqa: E402


def main():
    dev, dt = torch.device("cuda:0"), torch.float16
    with torch.device(dev):
        vae = AutoencoderKL()
    vae.to(dtype=dt)
    vae.compute_dtype = dt
    z = torch.randn(8, 64, 64, 8, device=dev).to(dt)
    z[..., 4:] = 0
    img = torch.rand(8, 512, 512, 8, device=dev).to(dt)
    img[..., 3:] = 0
    for _ in range(3):
        vae.decode_tokens(z)
        if "--encode" in sys.argv:
            vae.encode_tokens(img)
    torch.cuda.synchronize()
    t = threading.Timer(60.0, os._exit, [0])
    t.daemon = True
    t.start()


if __name__ == "__main__":
    main()
