#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 5 90 python -c "import __graft_entry__ as g; g.smoke()" > $O/r03_smoke_k.log 2>&1 || { echo "smoke failed or hung"; tail -3 $O/r03_smoke_k.log; exit 1; }
tail -1 $O/r03_smoke_k.log
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/r03_gputest_k.log 2>&1 || { echo "pytest failed"; tail -5 $O/r03_gputest_k.log; exit 1; }
tail -2 $O/r03_gputest_k.log
{
echo "== two-level chunk queue (base) vs one global counter (q1): 256 frames"
tools/exp_run.sh base q1 base q1
echo "== smaller chunks with the two-level queue: UHD frames 8 / 16 / 32 / 256, LUTR_CHUNK (tiles per chunk; default 8)"
for f in 8 16 32 256; do for n in base q1; do for ch in 8 4 2; do
  lib=lut_renderer_amd/lib/liblutr_$n.so; [ "$n" = base ] && lib=lut_renderer_amd/lib/liblutr.so
  LUTR_CHUNK=$ch LUTR_LIBRARY=$lib timeout -k 10 100 python bench.py --lean --no-stats --frames $f --variant vec_lds --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d.get('other_precision') or {}
print('frames %3d %-5s chunk %d  strict %6.1f (%7.1f us)  fast %6.1f' % ($f, '$n', $ch, d['value']/1e3, d['ms_per_step']*1e3, o.get('Mpx_s',0)/1e3))"
done; done; done
} > $O/r03_exp18.txt 2>&1
cat $O/r03_exp18.txt
