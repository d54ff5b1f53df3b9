#!/bin/bash
# whole-lattice mode (17^3): 12-byte strict nodes (base) vs float4 nodes (node16: one ds_read_b128 per tap)
O=gpurun_out; mkdir -p $O
{
echo "== 17^3 and 21^3 / 19^3 lattices whole in LDS, strict: 12-byte nodes (base) vs 16-byte nodes (node16), 64 UHD frames"
for lut in 17 19 21; do for dist in natural noise16 noise64 uniform; do for n in base; do
  lib=lut_renderer_amd/lib/liblutr_$n.so; [ "$n" = base ] && lib=lut_renderer_amd/lib/liblutr.so
  LUTR_LIBRARY=$lib timeout -k 10 100 python bench.py --lean --no-other --lut $lut --dist $dist --frames 64 --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%2d^3 %-8s %-6s strict %6.1f Gpx/s  %s' % ($lut, '$dist', '$n', d['value']/1e3, d['config']['kernel']))"
done; done; done
} > $O/r03_exp35.txt 2>&1
cat $O/r03_exp35.txt
