#!/bin/bash
# sweep the conv_halo_cc tile variants (per spatial size class) on a WRN-34-10 forward
set -e
N=${1:-2048}
for v in 0 1 2 3 4 5 6; do
  GRAPHITE_HALO_VARIANT="0:$v,1:$v,2:$v" timeout -k 10 120 python -u scripts/tune_wrn.py $N
done
GRAPHITE_CONV=igemm timeout -k 10 120 python -u scripts/tune_wrn.py $N
