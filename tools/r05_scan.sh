#!/bin/bash
# grid / stream scans of the current library on the 64-chain C3 batch (tests/gpu_profile.py): update blocks per batch, chain groups
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05_scan; mkdir -p $O
for b in 1536 2048 2560 3072 4096; do echo "upd_blocks $b: $(DA4ML_HIP_UPD_BLOCKS=$b timeout 90 python tests/gpu_profile.py 256 64 | head -1)"; done
for l in 2 3 4; do echo "lanes $l: $(DA4ML_HIP_LANES=$l timeout 90 python tests/gpu_profile.py 256 64 | head -1)"; done
for b in 2560 2560; do echo "repeat default: $(timeout 90 python tests/gpu_profile.py 256 64 | head -1)"; done
