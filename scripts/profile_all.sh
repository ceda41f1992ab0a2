#!/bin/bash
# usage: profile_all.sh <tag> <commit> [dtype ...]   (on the GPU box)  ->  gpurun_out/<tag>/
#        dtypes: bf16 f16 f32 f32x3 f32mix (default: all five); <commit> = `git rev-parse --short HEAD` of the tree that was pushed
#        (the box has no .git): it is stamped, with a hash of the kernel sources, into <file>.meta.json next to every counter file,
#        and bench.py reports it as roofline.counter_commit / roofline.stale.
# e.g.   gpurun --timeout 3000 -- "bash scripts/profile_all.sh r04a $(git rev-parse --short HEAD)"
# The rocprofv3 evidence behind each precision's `roofline` object of bench.py, from the SAME command with --dtype <dtype>:
#   <p>kernel_stats.csv, <p>last_forward.txt, <p>bench_profiled.json   rocprofv3 --kernel-trace --stats
#   <p>pmc_hbm_traffic.json        two separate --pmc passes (FETCH_SIZE / WRITE_SIZE)
#   <p>sq_counters.json            two separate --pmc passes of SQ counters, per kernel: mfma_busy_frac, lds_wait_frac,
#                                  lds_bank_conflict_frac, hbm_GBps = traffic / rocprof average duration
# with <p> = "" for bf16 and "<dtype>_" otherwise (bench.py's committed_counters() looks for profiles/r*_<p>....json).
# Every rocprofv3 command has the program directly behind `--` and uses --pmc without any trace option.
tag=$1
commit=$2
shift 2
dtypes="$@"; [ -z "$dtypes" ] && dtypes="bf16 f16 f32 f32x3 f32mix"
for dt in $dtypes; do
p=""; [ "$dt" != "bf16" ] && p="${dt}_"
steps=10; [ "$dt" = "f32" ] && steps=4
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# (--no-probes / --no-cups: nothing but the warm-up and the timed steps runs under the profiler, so per-kernel averages hold those steps only)
A="--dtype $dt --no-cpu-baseline --no-stream8 --no-cups --no-host-fed --no-probes --extra-dtypes="
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -- python3 $GRAFT_REPO_ROOT/bench.py --steps $steps --warmup 2 $A > $out/${p}bench_profiled.json 2> $out/${p}bench_profiled.err
echo "trace pass rc=$?"
stats=$(ls $out/raw/*/*kernel_stats.csv | head -1); trace=$(ls $out/raw/*/*kernel_trace.csv | head -1)
cp $stats $out/${p}kernel_stats.csv
python3 $GRAFT_REPO_ROOT/scripts/trace_summary.py $trace 100 > $out/${p}last_forward.txt
rm -rf $out/raw
timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 $A > $out/fetch.log 2>&1
echo "fetch pass rc=$?"
timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 $A > $out/write.log 2>&1
echo "write pass rc=$?"
timeout 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/sqa -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 $A > $out/sqa.log 2>&1
echo "sq pass A rc=$?"
timeout 500 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/sqb -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 $A > $out/sqb.log 2>&1
echo "sq pass B rc=$?"
cd $GRAFT_REPO_ROOT
python3 scripts/pmc_traffic.py $out/fetch $out/write > $out/${p}pmc_hbm_traffic.json
python3 scripts/pmc_sq_bench.py $out/sqa $out/sqb $out/${p}pmc_hbm_traffic.json $out/${p}kernel_stats.csv > $out/${p}sq_counters.json
rm -rf $out/fetch $out/write $out/sqa $out/sqb
for f in pmc_hbm_traffic sq_counters; do
  python3 -c "import json,sys,time; sys.path.insert(0,'.'); import bench; json.dump({'commit': '$commit', 'csrc_sha16': bench.csrc_sha16(), 'dtype': '$dt', 'collected_unix': int(time.time()), 'command': 'bench.py --dtype $dt --no-cpu-baseline --no-stream8 --no-cups --no-host-fed --no-probes --extra-dtypes='}, open('$out/${p}$f.meta.json','w'))"
done
tail -12 $out/${p}last_forward.txt
head -24 $out/${p}sq_counters.json
done
