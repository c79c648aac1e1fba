#!/bin/bash
O=$PWD/gpurun_out/r3pmc; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_]*LDS[A-Z_]*\|SQ_INSTS_[A-Z_]*\|SQ_ACTIVE_INST_[A-Z_]*\|SQ_INST_CYCLES_[A-Z_]*" | sort -u > $O/avail_sq.txt
(timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -o s -- python $R/tools/profile_forward.py > $O/pmc_sq.log 2>&1)
(timeout 300 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_lds -o l -- python $R/tools/profile_forward.py > $O/pmc_lds.log 2>&1)
cd $R
Q=$(find $O/pmc_sq -name "*counter_collection.csv" | head -1); python tools/pmc_family.py "$Q" > $O/r3_pmc_mfma_busy_by_family.txt 2>&1
Q=$(find $O/pmc_lds -name "*counter_collection.csv" | head -1); [ -n "$Q" ] && python tools/pmc_family.py "$Q" > $O/r3_pmc_lds_by_family.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
cat $O/avail_sq.txt | tr '\n' ' '; echo; cat $O/r3_pmc_mfma_busy_by_family.txt; cat $O/r3_pmc_lds_by_family.txt 2>/dev/null; tail -3 $O/pmc_lds.log
