# A/B of the warp kernels on the bench shard: quad-gather (default), one-lane-per-pixel gather, LDS band kernel
for V in "GRAPHITE_WARP_QUAD=1" "GRAPHITE_WARP_QUAD=0" "GRAPHITE_WARP_BAND=1"; do
  echo "== $V"
  env $V timeout -k 10 200 python bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline --candidates 512 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('images/s %.0f checksum %d warp ms/launch %.3f GB/s %.0f' % (d['value'], d['hits_checksum'], d['warp_roofline']['avg_launch_ms'], d['warp_roofline']['achieved']))" || exit 1
done
