#!/bin/bash
# usage: profile_pmc.sh <tag>  (on the GPU box): two separate counter passes of the bench command (3 steps each)
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/fetch.log 2>&1
echo "fetch pass rc=$?"
timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/write.log 2>&1
echo "write pass rc=$?"
cd $GRAFT_REPO_ROOT
python3 scripts/pmc_traffic.py $out/fetch $out/write > $out/pmc_hbm_traffic.json
rm -rf $out/fetch $out/write
head -40 $out/pmc_hbm_traffic.json
