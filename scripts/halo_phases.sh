# phase breakdown of the fp32 halo conv kernels (GRAPHITE_HALO_DBG bits: 1 no tile staging, 2 no MFMA loop, 4 no epilogue; results are wrong)
for d in ${1:-0 1 4 5 2}; do echo "== dbg $d: $(GRAPHITE_HALO_DBG=$d timeout -k 10 120 python scripts/tune_conv.py 32000 2>&1 | grep -o 'conv2_1.*conv4_2>[^;]*')"; done
