#!/bin/bash
# Sweep of the halo-kernel variants that run_gtsrb_halo (nets.hip) defines for every layer 1..5: 0 (default), 1, 2
# (tile shapes), 4 and 6 (A-prefetch pipelines).  Variant 3 exists for conv2_2 only and variant 7 (the small-launch
# tiling) for conv3_1 / conv4_1 only: they are swept per layer below so that no run asks for a missing variant.
set -e
for v in 0 1 2 4 6; do
  GRAPHITE_HALO_VARIANT="1:$v,2:$v,3:$v,4:$v,5:$v" timeout -k 10 120 python -u scripts/tune_conv.py 8192
done
GRAPHITE_HALO_VARIANT="1:3" timeout -k 10 120 python -u scripts/tune_conv.py 8192
GRAPHITE_HALO_VARIANT="2:7,4:7" timeout -k 10 120 python -u scripts/tune_conv.py 8192
