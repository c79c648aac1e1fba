#!/bin/bash
O=$PWD/gpurun_out/r3x; mkdir -p $O
(timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "norm or cast or layer" 2>&1 | tail -4) > $O/pytest_norm.log; cat $O/pytest_norm.log
for rnd in 1 2 3; do
for lib in libmimo_hip.so libmimo_hip_prev.so; do
  echo "== $lib" >> $O/ab.txt
  MIMO_HIP_LIB=$PWD/mimo_amd/$lib python - <<'PY' 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
import sys, torch
sys.path.insert(0, '.')
import bench
dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev, torch.float16)
t, fl, n, fam = bench.measure_forward(pipe, dev, torch.float16, 512, iters=6)
print(f"forward {t*1e3:.2f} ms", flush=True)
PY
done
done
cat $O/ab.txt
MIMO_HIP_LIB=$PWD/mimo_amd/libmimo_hip.so LN_CAPS=0 timeout 300 python tools/ln_micro.py 2>&1 | grep -v amdgpu.ids
