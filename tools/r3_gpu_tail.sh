#!/bin/bash
O=$PWD/gpurun_out/r3l; mkdir -p $O
(timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "ff_ or block_tail" 2>&1 | tail -15) > $O/pytest_ops.log
cat $O/pytest_ops.log
timeout 600 python tools/ff_trace.py 1 2 > $O/ff_trace.txt 2>&1
cat $O/ff_trace.txt | grep -v amdgpu.ids
