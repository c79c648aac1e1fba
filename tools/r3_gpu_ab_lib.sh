#!/bin/bash
# same-box A/B of several builds of the library: timings of the fused block tail kernels + the forward, interleaved
# usage: r3_gpu_ab_lib.sh <out dir name> <lib> <lib> ...
O=$PWD/gpurun_out/$1; mkdir -p $O; shift
for rnd in 1 2; do
for lib in "$@"; do
  echo "== $lib" >> $O/ab_lib.txt
  MIMO_HIP_LIB=$PWD/mimo_amd/$lib timeout 300 python tools/ff_trace.py --shipped 1 2 2>&1 | grep -v amdgpu.ids >> $O/ab_lib.txt
  MIMO_HIP_LIB=$PWD/mimo_amd/$lib python - <<'PY' 2>&1 | grep -v amdgpu.ids >> $O/ab_lib.txt
import sys, torch
sys.path.insert(0, '.')
import bench
from mimo_amd import ops
dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev, torch.float16)
t, fl, n, fam = bench.measure_forward(pipe, dev, torch.float16, 512, iters=5)
print(f"forward {t*1e3:.2f} ms, gemm family {fam['gemm_kernel']['ms']:.2f} ms over {fam['gemm_kernel']['launches']} launches", flush=True)
PY
done
done
cat $O/ab_lib.txt
