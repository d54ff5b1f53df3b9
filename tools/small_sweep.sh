#!/bin/bash
# tools/small_sweep.sh: where do the tile kernels overtake the vector kernels?  UHD and 1080p, 1..16 frames per launch.
for size in uhd 1080p; do for fmt in yuv420p10le yuv420p; do for f in 1 2 4 8 12 16 32; do
  a=$(timeout -k 10 100 python bench.py --lean --no-stats --no-other --size $size --fmt $fmt --frames $f --variant vec_global --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f' % (d['value']/1e3))")
  b=$(timeout -k 10 100 python bench.py --lean --no-stats --no-other --size $size --fmt $fmt --frames $f --variant vec_lds --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f' % (d['value']/1e3))")
  echo "$size $fmt frames $f: vec_global $a  vec_lds(tile2) $b Gpx/s"
done; done; done
