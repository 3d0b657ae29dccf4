for L in 0x70 0x71 0x73 0x77 0x7f; do echo "== links $L"; GRAPHITE_BF16_LINKS=$L timeout -k 10 120 python scripts/tune_bf16.py 32000 3 2>&1 | sed -n '/^bf16/,$p'; done
