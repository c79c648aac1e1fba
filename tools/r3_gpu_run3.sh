#!/bin/bash
# GPU run 3 of round 3: rocprofv3 statistics of the bench command (CSV), PMC traffic passes, CPU baseline at config-2 shapes,
# 768 bench with the fp8 record and the tiled VAE decode, full GPU test suite
set -x
O=$PWD/gpurun_out/r3c; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
(timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python $R/bench.py --no-cpu-baseline --no-bf16 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err)
(timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python $R/tools/profile_forward.py > $O/pmc_fetch.log 2>&1)
(timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python $R/tools/profile_forward.py > $O/pmc_write.log 2>&1)
(timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -o s -- python $R/tools/profile_forward.py > $O/pmc_sq.log 2>&1)
cd $R
find $O -name "*.csv" | head -30 > $O/csv_files.txt
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1); Q=$(find $O/pmc_sq -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py "$F" "$W" > $O/r3_pmc_forward_traffic.json 2> $O/pmc_traffic.err
python tools/pmc_family.py "$Q" > $O/r3_pmc_mfma_busy_by_family.txt 2>> $O/pmc_traffic.err
S=$(find $O/prof_bench -name "*kernel_stats.csv" | head -1); cp "$S" $O/r3_bench_kernel_stats_rocprofv3.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
(timeout 600 python bench.py --size 768 --fp8-qk --tile-vae 128 --no-cpu-baseline --no-bf16 > $O/bench_768.json 2> $O/bench_768.err)
(timeout 900 python bench.py --cpu-baseline-config2 --no-bf16 > $O/bench_cpu_config2.json 2> $O/bench_cpu_config2.err)
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15) > $O/pytest_all.log
cp gpurun_out/parity_report.txt $O/ 2>/dev/null
tail -3 $O/pytest_all.log; cat $O/csv_files.txt; du -sh $O
