#!/bin/bash
# RGB tube kernels: 12-byte nodes (read2_b32 + read_b32 per tap, 6 LDS cycles) vs 16-byte nodes (one ds_read_b128, 4 cycles; narrower tube)
O=gpurun_out; mkdir -p $O
{
echo "== k_rgb_tube 4-tap modes: 12-byte nodes (base) vs 16-byte nodes (n16), 128 UHD frames, strict"
for fmt in rgb24 rgba gbrp rgb48le gbrp10le; do for dist in natural vivid noise16; do for n in base n16; do
  lib=lut_renderer_amd/lib/liblutr_$n.so; [ "$n" = base ] && lib=lut_renderer_amd/lib/liblutr.so
  LUTR_RGB2=all LUTR_LIBRARY=$lib timeout -k 10 100 python bench.py --lean --no-other --fmt $fmt --frames 128 --dist $dist --variant vec_lds --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
t=d['config'].get('tiles') or d.get('tiles') or {}
print('%-9s %-8s %-5s %6.1f Gpx/s %.3f  %s %s' % ('$fmt', '$dist', '$n', d['value']/1e3, d['roofline']['frac'], d['config']['kernel'], d['config'].get('lds_window')))"
done; done; done
} > $O/r03_exp25.txt 2>&1
cat $O/r03_exp25.txt
