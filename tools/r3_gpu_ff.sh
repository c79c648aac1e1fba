#!/bin/bash
O=$PWD/gpurun_out/r3p; mkdir -p $O
(timeout 1500 python -m pytest tests/test_models_gpu.py -m gpu -q -k "sharded" 2>&1 | tail -12) > $O/pytest.log
tail -4 $O/pytest.log
