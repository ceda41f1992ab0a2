#!/bin/bash
# usage: prof_fwd.sh <tag> [env assignments...]  -> gpurun_out/<tag>/ kernel trace of scripts/probe_b64.py
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
env "$@" true
for kv in "$@"; do export "$kv"; done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag -- python3 $GRAFT_REPO_ROOT/scripts/probe_b64.py bf16 64 2>&1 | grep "ms/fwd"
