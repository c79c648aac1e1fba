#!/bin/bash
# same-box A/B of two builds of the library: shipped timings of the fused block tail kernels + the forward
O=$PWD/gpurun_out/r3q; mkdir -p $O
for rnd in 1 2; do
for lib in libmimo_hip.so libmimo_hip_prev.so; do
  echo "== $lib" >> $O/ab_lib.txt
  MIMO_HIP_LIB=$PWD/mimo_amd/$lib timeout 300 python tools/ff_trace.py --shipped 1 2 2>&1 | grep -v amdgpu.ids >> $O/ab_lib.txt
  MIMO_HIP_LIB=$PWD/mimo_amd/$lib python - <<'PY' 2>&1 | grep -v amdgpu.ids >> $O/ab_lib.txt
import sys, torch
sys.path.insert(0, '.')
import bench
from mimo_amd import ops
dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev, torch.float16)
for bt in (True, False):
    ops.BLOCK_TAIL_FUSED = bt
    t, fl, n, fam = bench.measure_forward(pipe, dev, torch.float16, 512, iters=5)
    print(f"BLOCK_TAIL_FUSED={bt}: forward {t*1e3:.2f} ms, gemm family {fam['gemm_kernel']['ms']:.2f} ms over {fam['gemm_kernel']['launches']} launches", flush=True)
PY
done
done
cat $O/ab_lib.txt
