#!/bin/bash
# RGB tube kernels: tile shape (lanes across x = 2^LW; a lane owns 16 / 8 / 4 pixels of one row)
O=gpurun_out; mkdir -p $O
{
echo "== k_rgb_tube tile shape: LUTR_LW_LOG2 = 5 (32 x 2 lanes, default) / 4 (16 x 4) / 3 (8 x 8), 128 UHD frames, strict"
for fmt in rgb24 rgba gbrp gbrp10le rgba64le; do for dist in natural vivid noise16; do for lw in 5 4 3; do
  LUTR_LW_LOG2=$lw LUTR_RGB2=all timeout -k 10 100 python bench.py --lean --no-other --fmt $fmt --frames 128 --dist $dist --variant vec_lds --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
w=d['config'].get('lds_window') or {}
print('%-9s %-8s lw %d %6.1f Gpx/s %.3f  %s tube %s gather %s of %s' % ('$fmt', '$dist', $lw, d['value']/1e3, d['roofline']['frac'], d['config']['kernel'], w.get('tube_tiles'), w.get('global_tiles'), w.get('tiles')))"
done; done; done
} > $O/r03_exp27.txt 2>&1
cat $O/r03_exp27.txt
