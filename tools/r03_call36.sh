#!/bin/bash
# RGB tube kernels, whole-lattice mode (17^3): float4 nodes (default) vs 12-byte nodes (LUTR_NO_WHOLE16=1)
O=gpurun_out; mkdir -p $O
{
echo "== k_rgb_tube, 17^3 lattice whole in LDS: 16-byte nodes (default) vs 12-byte (LUTR_NO_WHOLE16=1), 64 UHD frames, strict Gpx/s"
for fmt in rgb24 gbrp rgba gbrp10le; do for dist in natural noise16 uniform; do for w in 16 12; do
  env=""; [ $w = 12 ] && env="LUTR_NO_WHOLE16=1"
  env $env LUTR_RGB2=all timeout -k 10 100 python bench.py --lean --no-other --fmt $fmt --lut 17 --dist $dist --frames 64 --variant vec_lds --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-9s %-8s nodes %2d B  %6.1f Gpx/s %.3f  %s' % ('$fmt', '$dist', $w, d['value']/1e3, d['roofline']['frac'], d['config']['kernel']))"
done; done; done
} > $O/r03_exp36.txt 2>&1
cat $O/r03_exp36.txt
