#!/bin/bash
set -e
for v in 0 4 5 6 7; do
  GRAPHITE_HALO_VARIANT="1:$v,2:$v,3:$v,4:$v,5:$v" timeout -k 10 120 python -u scripts/tune_conv.py 8192
done
