"""Runs the level-0 spatial attention shape a few times (for rocprofv3 --pmc passes).  python tools/attn_only.py [--slow]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
n, N, C = 48, 4096, 320
qkv = torch.randn(n, N, 3 * C, device=dev).half()
bank = torch.randn(N, 2 * C, device=dev).half()
q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
for _ in range(3):
    ops.attention(q, k, v, 8, k2=bank[:, :C], v2=bank[:, C:], seg2_first_batch=24, q_prescaled="--slow" not in sys.argv)
torch.cuda.synchronize()
import threading
t = threading.Timer(30.0, os._exit, [0])
t.daemon = True
t.start()
