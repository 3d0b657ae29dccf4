#!/bin/bash
# runs scripts/tune_conv.py under a list of variant settings (one process each)
set -e
for v in "0:0" "1:1" "1:2" "3:1" "5:1"; do
  GRAPHITE_HALO_VARIANT="$v" timeout -k 10 120 python -u scripts/tune_conv.py 8192
done
