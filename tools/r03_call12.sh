#!/bin/bash
O=gpurun_out; mkdir -p $O
{
echo "== tube width against window size with mixed tiles on (64 frames; strict = value, fast = other)"
for cfg in "6 256 70" "7 128 85" "8 128 80" "9 128 90"; do set -- $cfg
  for d in natural vivid noise16; do
    LUTR_TUBE_H=$1 LUTR_MIN_WIN=$2 LUTR_TUBE_PCT=$3 timeout -k 10 100 python bench.py --lean --dist $d --frames 64 --steps 30 --warmup 8 2>$O/err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); w=d['config'].get('lds_window') or {}; o=d.get('other_precision') or {}
print('H<=$1 minwin=$2 pct=$3 %-8s strict %6.1f (tube %s mixed %s gather %s)  fast %6.1f  of %s tiles' % ('$d', d['value']/1e3, w.get('tube_tiles'), w.get('mixed_tiles'), w.get('global_tiles'), o.get('Mpx_s',0)/1e3, w.get('tiles')))"
  done
done
} > $O/r03_exp12.txt 2>&1
cat $O/r03_exp12.txt
