#!/bin/bash
O=$PWD/gpurun_out/r3attn; mkdir -p $O
(timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "attention or attn" 2>&1 | tail -4) > $O/pytest_attn.log; cat $O/pytest_attn.log
for rnd in 1 2 3; do
for lib in libmimo_hip.so libmimo_hip_noperm.so; do
  echo "== $lib" >> $O/ab.txt
  MIMO_HIP_LIB=$PWD/mimo_amd/$lib python - <<'PY' 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
import sys, torch
sys.path.insert(0, '.')
import bench
from mimo_amd import ops
dev = torch.device("cuda:0")
# level-0 spatial attention in isolation: 48 images, 4096 queries, 8 heads of d = 40, bank segment for the cond half
g = torch.Generator(device="cpu").manual_seed(1)
qkv = (torch.randn(48, 4096, 960, generator=g) * 0.5).to(dev).half()
bank = (torch.randn(4096, 640, generator=g) * 0.5).to(dev).half()
q, k, v = qkv[..., :320], qkv[..., 320:640], qkv[..., 640:]
fn = lambda: ops.attention(q, k, v, 8, k2=bank[:, :320], v2=bank[:, 320:], seg2_first_batch=24, q_prescaled=True)
for _ in range(3): fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(10): fn()
e1.record(); torch.cuda.synchronize()
print(f"attn40 level 0 (48 x 4096 queries, self + bank): {e0.elapsed_time(e1)/10:.3f} ms", flush=True)
pipe = bench.build_pipeline(dev, torch.float16)
t, fl, n, fam = bench.measure_forward(pipe, dev, torch.float16, 512, iters=6)
print(f"forward {t*1e3:.2f} ms, attention family {fam['attn_kernel']['ms']:.2f} ms", flush=True)
PY
done
done
cat $O/ab.txt
