"""Runs the level-0 spatial attention shape a few times (for rocprofv3 --pmc passes).  python tools/attn_only.py [--slow]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if any(k.startswith("MIMO_ATTN40_") for k in os.environ) or "--time" in sys.argv:  # variant knobs exist in the tune build only
    os.environ.setdefault("MIMO_HIP_LIB", os.path.join(ROOT, "mimo_amd", "libmimo_hip_tune.so"))
from mimo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
n, N, C = 48, 4096, 320
qkv = torch.randn(n, N, 3 * C, device=dev).half()
bank = torch.randn(N, 2 * C, device=dev).half()
q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
def run():
    return ops.attention(q, k, v, 8, k2=bank[:, :C], v2=bank[:, C:], seg2_first_batch=24, q_prescaled="--slow" not in sys.argv)


for _ in range(3):
    run()
torch.cuda.synchronize()
if "--time" in sys.argv:  # A/B of an environment knob read per launch: python tools/attn_only.py --time MIMO_ATTN_KT32
    knob = sys.argv[sys.argv.index("--time") + 1]
    i = sys.argv.index("--time") + 2
    vals = sys.argv[i].split(",") if len(sys.argv) > i and not sys.argv[i].startswith("--") else ["0", "1"]
    ref = None
    for env in vals + vals:
        os.environ[knob] = env
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        run()
        st.record()
        for _ in range(10):
            o = run()
        en.record()
        torch.cuda.synchronize()
        if ref is None:
            ref = o.float().clone()
        print(f"{knob}={env}: {st.elapsed_time(en)/10:.3f} ms  checksum {float(o.float().abs().mean()):.6f}  "
              f"rel-L2 vs the first variant {float((o.float() - ref).norm() / ref.norm()):.2e}", flush=True)
import threading
t = threading.Timer(30.0, os._exit, [0])
t.daemon = True
t.start()
