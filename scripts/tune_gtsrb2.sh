#!/bin/bash
set -e
for v in "0:0,1:0,2:0,3:0,4:0,5:0" "0:1,1:4,2:4,3:4,4:4,5:4" "0:1,1:5,2:5,3:5,4:5,5:5"; do
  GRAPHITE_HALO_VARIANT="$v" timeout -k 10 120 python -u scripts/tune_conv.py 8192
done
