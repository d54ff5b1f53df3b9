#!/bin/bash
# RGB tube kernels: coordinate reads of 4 pixels in flight (gp4) instead of 2 (base) in the 12-word layouts (planar, rgb24)
O=gpurun_out; mkdir -p $O
{
echo "== k_rgb_tube: 4-pixel coordinate groups (gp4) vs 2 (base), 128 UHD frames, strict Gpx/s, two rounds"
for rep in 1 2; do for fmt in gbrp10le gbrp rgb24; do for dist in natural noise16; do for n in base gp4; do
  lib=lut_renderer_amd/lib/liblutr_$n.so; [ "$n" = base ] && lib=lut_renderer_amd/lib/liblutr.so
  LUTR_RGB2=all LUTR_LIBRARY=$lib timeout -k 10 100 python bench.py --lean --no-other --fmt $fmt --dist $dist --frames 128 --variant vec_lds --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-9s %-8s %-5s %6.1f Gpx/s %.3f  %s' % ('$fmt', '$dist', '$n', d['value']/1e3, d['roofline']['frac'], d['config']['kernel']))"
done; done; done; done
} > $O/r03_exp43.txt 2>&1
cat $O/r03_exp43.txt
