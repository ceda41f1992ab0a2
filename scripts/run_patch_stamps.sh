#!/bin/bash
# per-K-step shader-clock stamps of the patch-resident kernels (needs lib/libokp_hip_S.so: OKP_EXTRA_CFLAGS=-DOKP_PATCH_STAMPS build)
export OKP_LIB=$GRAFT_REPO_ROOT/object_keypoints_amd/lib/libokp_hip_S.so
for a in "$@"; do
  echo "=== $a"; python3 scripts/patch_stamps.py $a 2>&1 | grep -v amdgpu.ids
done
