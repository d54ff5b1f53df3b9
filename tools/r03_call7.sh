#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r03_gputest_g.log 2>&1; echo "pytest rc=$?" | tee -a $O/r03_gputest_g.log
tail -3 $O/r03_gputest_g.log | cut -c1-300
timeout -k 10 200 python tools/soak.py 40 > $O/r03_soak_g.txt 2>&1; tail -1 $O/r03_soak_g.txt
{
echo "== mixed tiles (base) vs all-or-nothing vote (nomix)"
tools/exp_run.sh base nomix
tools/ab_dist.sh base nomix
echo "== mix_max sweep, noise16 and vivid, 64 frames"
for mm in 4 8 16 32 63; do for d in noise16 vivid; do
  LUTR_MIX_MAX=$mm timeout -k 10 100 python bench.py --lean --dist $d --frames 64 --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); w=d['config'].get('lds_window') or {}; o=d.get('other_precision') or {}
print('mix_max $mm %-8s strict %6.1f fast %6.1f  tube %s mixed %s gather %s of %s' % ('$d', d['value']/1e3, o.get('Mpx_s',0)/1e3, w.get('tube_tiles'), w.get('mixed_tiles'), w.get('global_tiles'), w.get('tiles')))"
done; done
} > $O/r03_exp7.txt 2>&1
cat $O/r03_exp7.txt
