#!/bin/bash
# RGB tube kernels: gather body that skips the pixel groups the optimistic pass got right (base) vs the full gather body (noskip)
O=gpurun_out; mkdir -p $O
{
echo "== k_rgb_tube: per-group skip in the gather body (base) vs full gather body (noskip), 128 UHD frames, strict"
for fmt in rgb24 rgba gbrp rgb48le gbrp10le; do for dist in natural vivid noise8 noise16; do for n in base noskip; do
  lib=lut_renderer_amd/lib/liblutr_$n.so; [ "$n" = base ] && lib=lut_renderer_amd/lib/liblutr.so
  LUTR_RGB2=all LUTR_LIBRARY=$lib timeout -k 10 100 python bench.py --lean --no-other --fmt $fmt --frames 128 --dist $dist --variant vec_lds --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
w=d['config'].get('lds_window') or {}
print('%-9s %-8s %-6s %6.1f Gpx/s %.3f  %s tube %s gather %s of %s' % ('$fmt', '$dist', '$n', d['value']/1e3, d['roofline']['frac'], d['config']['kernel'], w.get('tube_tiles'), w.get('global_tiles'), w.get('tiles')))"
done; done; done
} > $O/r03_exp26.txt 2>&1
cat $O/r03_exp26.txt
