#!/bin/bash
# Developer probe: tools/gpu_ab.py's columns for every variant under several AMC_BLOCKS_PER_CU settings (same box)
cd "$(dirname "$0")/.."
for bpc in 4 5 6 8; do echo "AMC_BLOCKS_PER_CU=$bpc"; AMC_BLOCKS_PER_CU=$bpc python3 tools/gpu_ab.py run 1 | tail -n +2; done
