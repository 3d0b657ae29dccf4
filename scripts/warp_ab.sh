# A/B of the warp kernels on the bench shard: band kernel on / off
for B in 1 0; do
  echo "== GRAPHITE_WARP_BAND=$B"
  GRAPHITE_WARP_BAND=$B timeout -k 10 200 python bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline --candidates 512 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('images/s %.0f checksum %d warp ms/launch %.3f GB/s %.0f' % (d['value'], d['hits_checksum'], d['warp_roofline']['avg_launch_ms'], d['warp_roofline']['achieved']))"
done
