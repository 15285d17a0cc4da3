#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -k "golden_and_twin or longer_horizons_match or row_layout_boundaries or other_horizons_matches or unclipped_long" 2>&1 | grep -v "^    \|^$" | tail -150 > gpurun_out/r3_pytest7.log
cat gpurun_out/r3_pytest7.log
