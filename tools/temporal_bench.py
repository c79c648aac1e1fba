"""Temporal (motion-module) attention at the three UNet levels: time and bytes / time (GPU box).
    python tools/temporal_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimo_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    for HW, C in [(4096, 320), (1024, 640), (256, 1280), (64, 1280)]:
        M = 2 * 24 * HW
        pool = [torch.randn(M, 3 * C, device=dev).to(torch.float16) for _ in range(max(2, int(600e6 // (M * 3 * C * 2)) + 1))]
        it = [0]

        def run():
            it[0] += 1
            qkv = pool[it[0] % len(pool)]
            return ops.temporal_attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], 2, 24, HW, 8)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(10):
                run()
            en.record()
            torch.cuda.synchronize()
            best = min(best, st.elapsed_time(en) / 10 * 1e-3)
        by = M * 4 * C * 2
        print(f"temporal attention HW{HW} C{C} (d = {C // 8}): {best*1e6:8.1f} us  {by/best/1e12:5.2f} TB/s (cold inputs)", flush=True)


if __name__ == "__main__":
    main()
