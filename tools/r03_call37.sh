#!/bin/bash
# packed fp32 (v_pk_add/mul_f32) in parts of the strict body: bit mask 1 luma add, 2 blends, 4 * M, 8 chroma sums
O=gpurun_out; mkdir -p $O
{
echo "== strict headline kernel with packed fp32 in parts of the body (LUTR_T2_PK bit mask), 256 UHD frames, two rounds"
for rep in 1 2; do for n in base pk1 pk2 pk4 pk8 pk5 pk15; do
  lib=lut_renderer_amd/lib/liblutr_$n.so; [ "$n" = base ] && lib=lut_renderer_amd/lib/liblutr.so
  LUTR_LIBRARY=$lib timeout -k 10 100 python bench.py --lean --no-other --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-5s strict %6.1f Gpx/s  %.4f' % ('$n', d['value']/1e3, d['roofline']['frac']))"
done; done
} > $O/r03_exp37.txt 2>&1
cat $O/r03_exp37.txt
