#!/bin/bash
# GPU run 4: in-situ per-dispatch timeline of the denoising forward (kernel trace kept), to compare with isolated microbenchmarks
set -x
O=$PWD/gpurun_out/r3d; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/tools/profile_forward.py > $O/trace.log 2>&1)
cd $R
cp $O/trace/t_kernel_trace.csv $O/forward_kernel_trace.csv 2>/dev/null || cp $(find $O/trace -name "*kernel_trace.csv" | head -1) $O/forward_kernel_trace.csv
rm -rf $O/trace
timeout 600 python tools/microbench.py --ab ";MIMO_GEMM_CFG=1;MIMO_GEMM_CFG=2" > $O/microbench_cfg.txt 2>&1
ls -la $O; head -3 $O/forward_kernel_trace.csv
