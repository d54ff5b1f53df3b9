#!/bin/bash
# tools/ab_small.sh LIB...: the vector kernels (small launches, packed RGB) under several libraries
for a in "--fmt yuv420p10le --frames 4 --variant vec_global" "--fmt yuv420p --size 1080p --frames 8 --variant vec_global" "--fmt rgb24 --frames 32" "--fmt rgba64le --frames 16" "--fmt gbrp10le --frames 4 --variant vec_global"; do for n in "$@"; do
  lib=lut_renderer_amd/lib/liblutr_$n.so; [ "$n" = base ] && lib=lut_renderer_amd/lib/liblutr.so
  LUTR_LIBRARY=$lib timeout -k 10 100 python bench.py --lean --no-stats --no-other $a --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-60s %-5s %6.1f Gpx/s %5.0f GB/s %s' % ('$a', '$n', d['value']/1e3, d['roofline']['achieved'], d['config']['kernel']))"
done; done
