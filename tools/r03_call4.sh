#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/r03_gputest_d.log 2>&1; echo "pytest rc=$?" | tee -a $O/r03_gputest_d.log
tail -3 $O/r03_gputest_d.log
timeout -k 10 200 python tools/soak.py 45 > $O/r03_soak_d.txt 2>&1; tail -2 $O/r03_soak_d.txt
{
echo "== windows in (g-r, b-g) too (base) vs windows in (g-r, b-r) with the b-g... no: wbr = everything b-r"
tools/exp_run.sh base wbr
tools/ab_dist.sh base wbr
echo "== strict tube width vs window size (base)"
for cfg in "6 256 70" "7 128 85"; do set -- $cfg
  for d in natural noise8 noise16 vivid; do
    LUTR_TUBE_H=$1 LUTR_MIN_WIN=$2 LUTR_TUBE_PCT=$3 timeout -k 10 100 python bench.py --lean --no-other --dist $d --frames 64 --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); w=d['config'].get('lds_window') or {}
print('H=$1 minwin=$2 %-8s strict %6.1f Gpx/s  tube %s level2 %s restage %s gather %s of %s tiles' % ('$d', d['value']/1e3, w.get('tube_tiles'), w.get('level2_tiles'), w.get('misses'), w.get('global_tiles'), w.get('tiles')))"
  done
done
} > $O/r03_exp4.txt 2>&1
cat $O/r03_exp4.txt
bash tools/pmc_units.sh > $O/r03_pmc_units_strict.txt 2>&1; cat $O/r03_pmc_units_strict.txt | tail -20
