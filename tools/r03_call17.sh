#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_fast_variant.py -m gpu -x -q > $O/r03_gputest_j.log 2>&1; echo "pytest rc=$?"; tail -2 $O/r03_gputest_j.log
{
echo "== static share of the chunk queue (LUTR_STATIC_PCT; 0 = round 2's first-chunk-only), 256 frames, strict / fast"
for pct in 0 50 85 95 100 0 85; do
  LUTR_STATIC_PCT=$pct timeout -k 10 100 python bench.py --lean --no-stats 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d.get('other_precision') or {}
print('static %3d %%  strict %6.1f  fast %6.1f' % ($pct, d['value']/1e3, o.get('Mpx_s',0)/1e3))"
done
echo "== 64 frames, content kinds"
for pct in 0 85; do for d in natural vivid noise16; do
  LUTR_STATIC_PCT=$pct timeout -k 10 100 python bench.py --lean --no-stats --dist $d --frames 64 --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d.get('other_precision') or {}
print('static %3d %% %-8s strict %6.1f  fast %6.1f' % ($pct, '$d', d['value']/1e3, o.get('Mpx_s',0)/1e3))"
done; done
} > $O/r03_exp17.txt 2>&1
cat $O/r03_exp17.txt
