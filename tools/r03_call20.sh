#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 5 90 python -c "import __graft_entry__ as g; g.smoke()" > $O/r03_smoke_m.log 2>&1 || { echo "smoke failed or hung"; tail -3 $O/r03_smoke_m.log; exit 1; }
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/r03_gputest_m.log 2>&1 || { echo "pytest failed"; tail -15 $O/r03_gputest_m.log; exit 1; }
tail -2 $O/r03_gputest_m.log
{
echo "== RGB tube kernels with the two-level queue: frames per launch"
for cfg in "rgb24 8" "rgb24 32" "rgb24 128" "rgba 8" "rgba 128" "gbrp 8" "gbrp 128" "rgb48le 64" "rgba64le 64"; do set -- $cfg
  for v in vec_global vec_lds; do
  timeout -k 10 100 python bench.py --lean --no-other --fmt $1 --frames $2 --variant $v --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-9s frames %3d %-10s %6.1f Gpx/s (%7.1f us) %.3f  %s' % ('$1', $2, '$v', d['value']/1e3, d['ms_per_step']*1e3, d['roofline']['frac'], d['config']['kernel']))"
done; done
} > $O/r03_exp20.txt 2>&1
cat $O/r03_exp20.txt
