#!/bin/bash
# tools/ab_fmt.sh LIB...: memory-side formats under several libraries
for a in "--fmt gbrp10le --frames 128" "--fmt yuv444p10le --frames 128" "--fmt yuv422p10le" "--fmt yuv420p10le"; do for n in "$@"; do
  lib=lut_renderer_amd/lib/liblutr_$n.so; [ "$n" = base ] && lib=lut_renderer_amd/lib/liblutr.so
  LUTR_LIBRARY=$lib timeout -k 10 100 python bench.py --lean --no-stats $a --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d.get('other_precision') or {}
print('%-34s %-5s %6.1f Gpx/s %5.0f GB/s  (fast %6.1f)' % ('$a', '$n', d['value']/1e3, d['roofline']['achieved'], o.get('Mpx_s',0)/1e3))"
done; done
