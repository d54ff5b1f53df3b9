#!/bin/bash
# frames per launch: the default 256 against deeper batches (the launch's fixed cost amortises; 288 GB of HBM holds 2048 UHD frames in + out)
O=gpurun_out; mkdir -p $O
{
echo "== UHD yuv420p10le tetrahedral strict | fast Gpx/s by frames per launch, two rounds"
for rep in 1 2; do for f in 256 512 1024 2048; do
  timeout -k 10 200 python bench.py --lean --frames $f --steps 20 --warmup 6 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d.get('other_precision') or {}
print('frames %4d  strict %6.1f Gpx/s (%.4f)  fast %6.1f  setup %.2f s' % ($f, d['value']/1e3, d['roofline']['frac'], o.get('Mpx_s',0)/1e3, d['config']['setup_s']))"
done; done
} > $O/r03_exp38.txt 2>&1
cat $O/r03_exp38.txt
