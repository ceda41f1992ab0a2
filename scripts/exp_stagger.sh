#!/bin/bash
echo "--- patch 64x64"; for s in "" "2,6" "2,12" "2,20" "4,6" "4,10" ""; do echo "stagger=$s: $(OKP_PATCH_STAGGER=$s python3 scripts/probe_patch.py 2>&1 | grep 'tile 13' | tr '\n' ' ')"; done
echo "--- patch 128x128"; for s in "" "2,6" "2,12" "2,20" "4,6" "4,10" ""; do echo "stagger=$s: $(OKP_PATCH_STAGGER=$s python3 scripts/probe_patch.py hw=128 2>&1 | grep 'tile 13' | tr '\n' ' ')"; done
echo "--- patch 128x128 s2 cin 128"; for s in "" "2,6" "2,12" "2,20" "4,6" "4,10" ""; do echo "stagger=$s: $(OKP_PATCH_STAGGER=$s python3 scripts/probe_patch.py hw=128 cin=128 stride=2 2>&1 | grep 'tile 13' | tr '\n' ' ')"; done
echo "--- unpool"; for s in "" "2,6" "2,12" "2,20" "4,6" "4,10" ""; do echo "stagger=$s: $(OKP_PATCH_STAGGER=$s python3 scripts/probe_unpool.py 2>&1 | grep -v amdgpu | tail -2 | tr '\n' ' ')"; done
