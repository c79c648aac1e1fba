#!/bin/bash
# per-shape table of the forward's GEMM / conv launches against their own ceilings + in-situ timeline of the current build
O=$PWD/gpurun_out/r3k; mkdir -p $O
R=$PWD
timeout 400 python tools/forward_bound.py --size 512 > $O/forward_bound_shapes_512.txt 2>&1
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/tools/profile_forward.py > $O/trace.log 2>&1)
cd $R
cp $(find $O/trace -name "*kernel_trace.csv" | head -1) $O/forward_kernel_trace.csv
rm -rf $O/trace
python tools/trace_forward.py $O/forward_kernel_trace.csv > $O/forward_timeline.txt 2>&1
rm -f $O/forward_kernel_trace.csv
cat $O/forward_bound_shapes_512.txt | tail -70
