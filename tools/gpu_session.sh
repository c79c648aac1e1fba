#!/bin/bash
# One parameterised GPU-box session (replaces the per-run r3_gpu_*.sh scripts).  Run through gpurun from the repo root:
#   gpurun --timeout 1500 -- 'bash tools/gpu_session.sh <tag> <stage> [<stage> ...]'
# Outputs go to gpurun_out/<tag>/ (merged back by gpurun); copy what is to be judged into profiles/.
# Stages:
#   tests            full GPU test tier                        -> pytest_gpu.txt (+ parity_report.txt)
#   tests:<expr>     pytest -k <expr>                          -> pytest_<n>.txt
#   bench            default bench.py line                     -> bench_default.json
#   bench768         configs[4]: --size 768 --fp8-qk --tile-vae 128
#   bench_edit       default line + the configs[2] `edit` sub-record
#   bench_rccl1      torchrun --nproc-per-node 1 ... --shard-windows --force-shard (the RCCL path at world 1)
#   prof             rocprofv3 --kernel-trace --stats of the bench command -> bench_kernel_stats_rocprofv3.csv + JSON line
#   pmc_traffic      FETCH_SIZE / WRITE_SIZE passes (separate) over tools/profile_forward.py -> pmc_forward_traffic.json
#   pmc_sq           MFMA-busy / wave-state counters by kernel family  -> pmc_mfma_busy_by_family.txt
#   pmc_clock        GRBM_GUI_ACTIVE / duration per kernel family = the effective (power-limited) shader clock
#   pmc_lds          LDS bank-conflict counters by kernel family       -> pmc_lds_by_family.txt
#   timeline         in-situ per-dispatch timeline of one forward      -> forward_timeline.txt
#   bound            per-shape ceilings of the forward's launches      -> forward_bound_shapes_512.txt
#   hconv            tools/hconv_bench.py (old two-launch path vs fused conv)
#   hconv_variants   tools/hconv_variants.py (needs `python tools/hconv_variants.py --build` before the gpurun call)
#   vae_bound        tools/vae_bound.py --frames 8
#   attn_ab:<KNOB> <v0,v1,..>   tools/attn_only.py --time (level-0 spatial attention under a tune-library knob)
#   envtests:<ENV=..>;<expr>    pytest -k <expr> on the tune library with environment knobs set
#   emulate8         bench.py --emulate-world 8 (per-rank schedule of the 8-GPU long-clip job measured on one GPU)
#   ceiling          tools/ceiling.py (MFMA-only probe + best-case 8192^3 GEMMs)
#   fillprobe        tools/_ab/lds_fill_probe (build it first: see tools/probes/lds_fill_probe.hip) -> lds_fill_probe.txt
#   bound:<args>     tools/forward_bound.py --size 512 <args> (e.g. "bound:--mfma-rate 1.25 --attn-rate 1.0")
#   sh:<command>     any shell command (output -> sh_<n>.txt)
#   fftrace:<modes>  tools/ff_trace.py <modes> (phase trace + ablations of the fused tail / head, tune library)
#   head             tools/head_bench.py (fused block head vs the launches it replaces)
#   ab:<settings>    tools/ab_forward.py <settings> (space-separated ops knobs, e.g. "ab:BLOCK_HEAD_FUSED=0 BLOCK_HEAD_FUSED=1")
set -x
TAG=$1; shift
R=$PWD
O=$R/gpurun_out/$TAG; mkdir -p $O
clean() { find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete; }
n=0
for stage in "$@"; do
  n=$((n + 1))
  case $stage in
    tests) (timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15) > $O/pytest_gpu.txt; cp gpurun_out/parity_report.txt $O/ 2>/dev/null; tail -3 $O/pytest_gpu.txt ;;
    tests:*) (timeout 1200 python -m pytest tests -m gpu -q -k "${stage#tests:}" 2>&1 | tail -15) > $O/pytest_$n.txt; tail -3 $O/pytest_$n.txt ;;
    bench) (timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err); head -c 1500 $O/bench_default.json ;;
    bench768) (timeout 900 python bench.py --size 768 --fp8-qk --tile-vae 128 --no-cpu-baseline --no-bf16 > $O/bench_768.json 2> $O/bench_768.err); head -c 800 $O/bench_768.json ;;
    bench_edit) (timeout 900 python bench.py --edit --no-cpu-baseline --no-bf16 > $O/bench_edit.json 2> $O/bench_edit.err); head -c 400 $O/bench_edit.json ;;
    bench_rccl1) (timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 1 --warmup 1 --frames 48 --shard-windows --force-shard --no-cpu-baseline --no-bf16 2>&1 | tail -1) > $O/bench_rccl_world1.json; head -c 1200 $O/bench_rccl_world1.json ;;
    prof) cd /tmp && export TMPDIR=/tmp
      (timeout 700 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python $R/bench.py --no-cpu-baseline --no-bf16 > $O/bench_under_rocprofv3.json 2> $O/bench_under_rocprofv3.err)
      cd $R; S=$(find $O/prof_bench -name "*kernel_stats.csv" | head -1); cp "$S" $O/bench_kernel_stats_rocprofv3.csv; rm -rf $O/prof_bench; head -12 $O/bench_kernel_stats_rocprofv3.csv ;;
    pmc_traffic) cd /tmp && export TMPDIR=/tmp
      (timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python $R/tools/profile_forward.py > $O/pmc_fetch.log 2>&1)
      (timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python $R/tools/profile_forward.py > $O/pmc_write.log 2>&1)
      cd $R; F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
      python tools/pmc_traffic.py "$F" "$W" > $O/pmc_forward_traffic.json 2> $O/pmc_traffic.err; rm -rf $O/pmc_fetch $O/pmc_write; cat $O/pmc_forward_traffic.json ;;
    pmc_sq) cd /tmp && export TMPDIR=/tmp
      (timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -o s -- python $R/tools/profile_forward.py > $O/pmc_sq.log 2>&1)
      cd $R; Q=$(find $O/pmc_sq -name "*counter_collection.csv" | head -1); python tools/pmc_family.py "$Q" > $O/pmc_mfma_busy_by_family.txt 2>&1; rm -rf $O/pmc_sq; cat $O/pmc_mfma_busy_by_family.txt ;;
    pmc_clock) cd /tmp && export TMPDIR=/tmp
      (timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_clk -o c -- python $R/tools/profile_forward.py > $O/pmc_clk.log 2>&1)
      cd $R; Q=$(find $O/pmc_clk -name "*counter_collection.csv" | head -1); T=$(find $O/pmc_clk -name "*kernel_trace.csv" | head -1)
      head -2 "$Q" > $O/pmc_clock_header.txt; (cd tools && python pmc_clock.py "$Q" "$T") > $O/pmc_clock_by_family.txt 2>&1; rm -rf $O/pmc_clk; cat $O/pmc_clock_by_family.txt ;;
    pmc_lds) cd /tmp && export TMPDIR=/tmp
      (timeout 300 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_lds -o l -- python $R/tools/profile_forward.py > $O/pmc_lds.log 2>&1)
      cd $R; Q=$(find $O/pmc_lds -name "*counter_collection.csv" | head -1); python tools/pmc_family.py "$Q" > $O/pmc_lds_by_family.txt 2>&1; rm -rf $O/pmc_lds; cat $O/pmc_lds_by_family.txt ;;
    timeline) cd /tmp && export TMPDIR=/tmp
      (timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/tools/profile_forward.py > $O/trace.log 2>&1)
      cd $R; T=$(find $O/trace -name "*kernel_trace.csv" | head -1); python tools/trace_forward.py "$T" > $O/forward_timeline.txt 2>&1; rm -rf $O/trace; head -40 $O/forward_timeline.txt ;;
    bound) timeout 400 python tools/forward_bound.py --size 512 2>&1 | grep -v amdgpu.ids > $O/forward_bound_shapes_512.txt; head -30 $O/forward_bound_shapes_512.txt ;;
    hconv) timeout 400 python tools/hconv_bench.py 2>&1 | grep -v amdgpu.ids > $O/hconv_bench.txt; cat $O/hconv_bench.txt ;;
    hconv_variants) timeout 500 python tools/hconv_variants.py 2>&1 | grep -v amdgpu.ids > $O/hconv_variants.txt; cat $O/hconv_variants.txt ;;
    attn_ab:*) timeout 300 python tools/attn_only.py --time ${stage#attn_ab:} 2>&1 | grep -v amdgpu.ids > $O/attn_ab_$n.txt; cat $O/attn_ab_$n.txt ;;
    envtests:*) (IFS=';' read -r ENVS EXPR <<< "${stage#envtests:}"; env MIMO_HIP_LIB=$R/mimo_amd/libmimo_hip_tune.so $ENVS timeout 900 python -m pytest tests -m gpu -q -k "$EXPR" 2>&1 | tail -8) > $O/pytest_env_$n.txt; tail -3 $O/pytest_env_$n.txt ;;
    emulate8) (timeout 600 python bench.py --emulate-world 8 > $O/bench_emulate_world8.json 2> $O/bench_emulate_world8.err); head -c 3000 $O/bench_emulate_world8.json ;;
    fillprobe) timeout 200 tools/_ab/lds_fill_probe > $O/lds_fill_probe.txt 2>&1; cat $O/lds_fill_probe.txt ;;
    ceiling) timeout 400 python tools/ceiling.py 2>&1 | grep -v amdgpu.ids > $O/mfma_ceiling.txt; cat $O/mfma_ceiling.txt ;;
    bound:*) timeout 400 python tools/forward_bound.py --size 512 ${stage#bound:} 2>&1 | grep -v amdgpu.ids > $O/forward_bound_$n.txt; head -8 $O/forward_bound_$n.txt ;;
    sh:*) (timeout 900 bash -c "${stage#sh:}" 2>&1 | grep -v amdgpu.ids) > $O/sh_$n.txt; cat $O/sh_$n.txt ;;
    fftrace:*) timeout 400 python tools/ff_trace.py ${stage#fftrace:} 2>&1 | grep -v amdgpu.ids > $O/ff_trace_$n.txt; cat $O/ff_trace_$n.txt ;;
    head) timeout 400 python tools/head_bench.py 2>&1 | grep -v amdgpu.ids > $O/head_bench.txt; cat $O/head_bench.txt ;;
    ab:*) timeout 600 python tools/ab_forward.py ${stage#ab:} 2>&1 | grep -v amdgpu.ids > $O/ab_forward_$n.txt; cat $O/ab_forward_$n.txt ;;
    vae_bound) timeout 600 python tools/vae_bound.py --frames 8 2>&1 | grep -v amdgpu.ids > $O/vae_bound.txt; cat $O/vae_bound.txt ;;
    *) echo "unknown stage $stage" ;;
  esac
  clean
done
du -sh $O
