#!/bin/bash
# A/B two builds in ONE gpurun call (devices differ by several % between calls): scripts/ab.sh <cmd...>
for r in 1 2; do
  for v in A B; do
    echo "== $v"; OKP_LIB=$GRAFT_REPO_ROOT/object_keypoints_amd/lib/libokp_hip_$v.so "$@" 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
