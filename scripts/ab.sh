#!/bin/bash
# A/B two builds in ONE gpurun call (devices differ by several % between calls; ABBA order cancels clock drift)
for v in A B B A; do
  {
    out=$(OKP_LIB=$GRAFT_REPO_ROOT/object_keypoints_amd/lib/libokp_hip_$v.so "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | sed -E 's/.*: ([0-9.]+ us\/launch.*)/\1/')
    echo "$v: $out"
  }
done
