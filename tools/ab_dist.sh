#!/bin/bash
# tools/ab_dist.sh LIBA LIBB: fast / strict Gpx/s of two libraries on natural, vivid, noise8, noise16 frames (64-frame launches)
for d in natural vivid noise8 noise16; do for n in "$@"; do
  lib=lut_renderer_amd/lib/liblutr_$n.so; [ "$n" = base ] && lib=lut_renderer_amd/lib/liblutr.so
  LUTR_LIBRARY=$lib timeout -k 10 100 python bench.py --lean --dist $d --frames 64 --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d.get('other_precision') or {}; w=d['config'].get('lds_window') or {}
print('%-8s %-8s strict %6.1f fast %6.1f  tube %s / %s tiles (strict)' % ('$d', '$n', d['value']/1e3, o.get('Mpx_s',0)/1e3, w.get('tube_tiles'), w.get('tiles')))"
done; done
