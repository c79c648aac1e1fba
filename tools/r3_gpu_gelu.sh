#!/bin/bash
O=$PWD/gpurun_out/r3o; mkdir -p $O
(timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "ff_ or block_tail or geglu" 2>&1 | tail -8) > $O/pytest_ops.log
cat $O/pytest_ops.log
timeout 300 python tools/ff_trace.py --shipped 1 2 2>&1 | grep -v amdgpu.ids > $O/ff_shipped.txt
cat $O/ff_shipped.txt
timeout 300 python tools/ff_trace.py 1 2 2>&1 | grep -v amdgpu.ids > $O/ff_trace_tail.txt
cat $O/ff_trace_tail.txt
python - <<'PY' > $O/ab.txt 2>&1
import sys, torch
sys.path.insert(0, '.')
import bench
from mimo_amd import ops
dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev, torch.float16)
for rnd in range(3):
    for bt in (False, True):
        ops.BLOCK_TAIL_FUSED = bt
        t, fl, n, fam = bench.measure_forward(pipe, dev, torch.float16, 512, iters=5)
        print(f"BLOCK_TAIL_FUSED={bt}: forward {t*1e3:.2f} ms, gemm family {fam['gemm_kernel']['ms']:.2f} ms over {fam['gemm_kernel']['launches']} launches", flush=True)
PY
grep -v amdgpu.ids $O/ab.txt
