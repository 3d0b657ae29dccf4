# sweep of the bf16 halo-kernel variants (production configuration: every link bf16), one layer at a time
for L in 1 2 3 4 5; do
  for v in 0 1 2 3 4; do
    out=$(GRAPHITE_BF16_VARIANT="$L:$v" timeout -k 10 120 python scripts/tune_bf16.py 32000 3 2>&1 | grep "conv_halo_bf16" | sed -n "${L}p")
    echo "layer $L variant $v: $out"
  done
done
