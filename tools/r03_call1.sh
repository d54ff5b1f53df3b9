#!/bin/bash
# round-3 GPU call 1: GPU test-suite on the new host code, the new default bench line, the 3+3-plane stream microbenchmark,
# and the first strict-kernel A/Bs (tube width against window size, 16-byte nodes).
O=gpurun_out; mkdir -p $O
which ffmpeg ffprobe > $O/r03_which_ffmpeg.txt 2>&1; echo "rc=$?" >> $O/r03_which_ffmpeg.txt
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/r03_gputest_a.log 2>&1; echo "pytest rc=$?" | tee -a $O/r03_gputest_a.log
tail -3 $O/r03_gputest_a.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/r03_bench_a.json 2> $O/r03_bench_a.err; echo "bench rc=$?"
tail -c 600 $O/r03_bench_a.json
timeout -k 10 200 ./tools/ubench/stream6 128 > $O/r03_stream6.txt 2>&1; echo "stream6 rc=$?"
{
echo "== libraries"; tools/exp_run.sh base n16
echo "== strict tube width vs window size (base library)"
for cfg in "6 256" "7 128" "7 144"; do set -- $cfg
  for d in natural noise16 vivid; do
    LUTR_TUBE_H=$1 LUTR_MIN_WIN=$2 LUTR_TUBE_PCT=80 timeout -k 10 100 python bench.py --lean --no-other --dist $d --frames 64 --steps 30 --warmup 8 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); w=d['config'].get('lds_window') or {}
print('H=$1 minwin=$2 %-8s strict %6.1f Gpx/s  tube %s level2 %s restage %s gather %s of %s tiles' % ('$d', d['value']/1e3, w.get('tube_tiles'), w.get('level2_tiles'), w.get('misses'), w.get('global_tiles'), w.get('tiles')))"
  done
done
} > $O/r03_exp1.txt 2>&1
cat $O/r03_exp1.txt
