#!/bin/bash
# round-3 closing run on one box: full GPU test tier, the default bench line (+ the same command under rocprofv3 --stats),
# PMC traffic passes stamped with the library hash, per-shape ceilings and the in-situ timeline of the final build
set -x
O=$PWD/gpurun_out/r3z; mkdir -p $O
R=$PWD
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15) > $O/pytest_all.log
cp gpurun_out/parity_report.txt $O/ 2>/dev/null
(timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err)
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python $R/bench.py --no-cpu-baseline --no-bf16 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err)
(timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python $R/tools/profile_forward.py > $O/pmc_fetch.log 2>&1)
(timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python $R/tools/profile_forward.py > $O/pmc_write.log 2>&1)
(timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/tools/profile_forward.py > $O/trace.log 2>&1)
cd $R
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py "$F" "$W" > $O/r3_pmc_forward_traffic.json 2> $O/pmc_traffic.err
S=$(find $O/prof_bench -name "*kernel_stats.csv" | head -1); cp "$S" $O/r3_bench_kernel_stats_rocprofv3.csv
T=$(find $O/trace -name "*kernel_trace.csv" | head -1); python tools/trace_forward.py "$T" > $O/forward_timeline.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
rm -rf $O/prof_bench $O/pmc_fetch $O/pmc_write $O/trace
timeout 400 python tools/forward_bound.py --size 512 2>&1 | grep -v amdgpu.ids > $O/forward_bound_shapes_512.txt
tail -3 $O/pytest_all.log; cat $O/bench_default.json | head -c 1500; du -sh $O
