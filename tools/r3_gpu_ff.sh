#!/bin/bash
O=$PWD/gpurun_out/r3j; mkdir -p $O
(timeout 900 python -m pytest tests/test_golden.py tests/test_models_gpu.py -m gpu -q -k "not world" 2>&1 | tail -12) > $O/pytest.log
python - <<'PY' > $O/ab.txt 2>&1
import sys, torch
sys.path.insert(0, '.')
import bench
from mimo_amd import ops
dev = torch.device("cuda:0")
pipe = bench.build_pipeline(dev, torch.float16)
for rnd in range(3):
    for ff, pj in ((True, True), (True, False), (False, False)):
        ops.FF_FUSED, ops.FF_PROJ_FUSED = ff, pj
        t, fl, n, fam = bench.measure_forward(pipe, dev, torch.float16, 512, iters=5)
        print(f"FF_FUSED={ff} FF_PROJ_FUSED={pj}: forward {t*1e3:.2f} ms, gemm family {fam['gemm_kernel']['ms']:.2f} ms over {fam['gemm_kernel']['launches']} launches", flush=True)
PY
tail -4 $O/pytest.log; cat $O/ab.txt; grep -n "config-2\|full-size\|multi-window\|sharded" gpurun_out/parity_report.txt | tail -8
