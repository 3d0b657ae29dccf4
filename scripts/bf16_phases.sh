# phase breakdown of the bf16 conv kernels (GRAPHITE_BF_DBG bits: 1 no tile staging, 2 no MFMA loop, 4 no epilogue; results are wrong)
for d in ${1:-0 1 2 4 3 5 6 7}; do echo "== dbg $d"; GRAPHITE_BF_DBG=$d timeout -k 10 120 python scripts/tune_bf16.py 32000 3 2>&1 | grep -A12 "^bf16" | grep "conv_halo_bf16\|^bf16" || exit 1; done
