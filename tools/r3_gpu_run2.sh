#!/bin/bash
# GPU run 2 of round 3: new kernel tests, rocprofv3 statistics of the bench command, PMC traffic, 768 bench with fp8 record
set -x
O=$PWD/gpurun_out/r3b; mkdir -p $O
R=$PWD
(timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "fp8 or d512 or temporal" 2>&1 | tail -15) > $O/pytest_new.log
(timeout 300 python -m pytest tests/test_golden.py -m gpu -q -k "vae_784" 2>&1 | tail -15) > $O/pytest_vae784.log
cd /tmp && export TMPDIR=/tmp
(timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench -- python $R/bench.py --no-cpu-baseline --no-bf16 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err)
(timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o f -- python $R/tools/profile_forward.py > $O/pmc_fetch.log 2>&1)
(timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o w -- python $R/tools/profile_forward.py > $O/pmc_write.log 2>&1)
cd $R
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py "$F" "$W" > $O/r3_pmc_forward_traffic.json 2> $O/pmc_traffic.err
S=$(find $O/prof_bench -name "*kernel_stats.csv" | head -1); cp "$S" $O/r3_bench_kernel_stats_rocprofv3.csv
# keep the merged output small: drop the raw traces
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
(timeout 600 python bench.py --size 768 --fp8-qk --no-cpu-baseline --no-bf16 > $O/bench_768.json 2> $O/bench_768.err)
(timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err)
tail -3 $O/pytest_new.log $O/pytest_vae784.log; head -c 600 $O/bench_default.json; du -sh $O
