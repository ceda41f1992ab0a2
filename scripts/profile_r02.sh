#!/bin/bash
# usage: profile_r02.sh <tag>   (on the GPU box)  ->  gpurun_out/<tag>/
#   bench.json                  the un-profiled default bench line (bf16 headline + fp32 / fp16 objects)
#   kernel_stats.csv, last_forward.txt, bench_profiled.json     rocprofv3 --kernel-trace --stats of the bench command
#   pmc_hbm_traffic.json        two separate --pmc passes (FETCH_SIZE / WRITE_SIZE) of the bench command
#   sq_counters.json            two separate --pmc passes of SQ counters of the bench command, per kernel:
#                               mfma_busy_frac, lds_wait_frac, lds_bank_conflict_frac, and hbm_GBps = traffic / rocprof average duration
# Every rocprofv3 command has the program directly behind `--` and uses --pmc without any trace option.
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --extra-dtypes ''"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --extra-dtypes "" > $out/bench_profiled.json 2> $out/bench_profiled.err
echo "trace pass rc=$?"
stats=$(ls $out/raw/*/*kernel_stats.csv | head -1); trace=$(ls $out/raw/*/*kernel_trace.csv | head -1)
cp $stats $out/kernel_stats.csv
python3 $GRAFT_REPO_ROOT/scripts/trace_summary.py $trace 100 > $out/last_forward.txt
rm -rf $out/raw
timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --extra-dtypes "" > $out/fetch.log 2>&1
echo "fetch pass rc=$?"
timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --extra-dtypes "" > $out/write.log 2>&1
echo "write pass rc=$?"
timeout 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/sqa -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --extra-dtypes "" > $out/sqa.log 2>&1
echo "sq pass A rc=$?"
timeout 500 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/sqb -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --extra-dtypes "" > $out/sqb.log 2>&1
echo "sq pass B rc=$?"
cd $GRAFT_REPO_ROOT
python3 scripts/pmc_traffic.py $out/fetch $out/write > $out/pmc_hbm_traffic.json
python3 scripts/pmc_sq_bench.py $out/sqa $out/sqb $out/pmc_hbm_traffic.json $out/kernel_stats.csv > $out/sq_counters.json
rm -rf $out/fetch $out/write $out/sqa $out/sqb
python3 bench.py --steps 100 --warmup 5 > $out/bench.json 2> $out/bench.err
tail -1 $out/bench.json | cut -c1-300
tail -14 $out/last_forward.txt
head -30 $out/sq_counters.json
