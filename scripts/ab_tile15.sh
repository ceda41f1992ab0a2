python -m pytest tests/test_gpu_conv.py -m gpu -x -q -k patch_resident 2>&1 | tail -5
for a in "hw=64" "hw=128" "hw=128 cin=128 stride=2" "hw=64 stride=2" "hw=64 res=1"; do echo "[$a] $(python3 scripts/probe_patch.py $a 2>&1 | grep tile | tr '\n' ' ')"; done
for a in "hw=32" "hw=16"; do for t in 13 15 15 13; do python3 scripts/probe_unpool.py $a tile=$t 2>&1 | tail -1; done; done
