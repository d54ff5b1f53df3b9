run() { timeout -k 10 100 python bench.py --lean --no-stats --no-other --fmt gbrp10le --frames 128 --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 %6.1f Gpx/s %5.0f GB/s' % (d['value']/1e3, d['roofline']['achieved']))"; }
run base
for l in 4 5 6; do LUTR_LW_LOG2=$l run "lw$l"; done
for c in 8 16 32; do LUTR_CHUNK=$c run "chunk$c"; done
for w in 12 16 20 24; do LUTR_WAVES_PER_CU=$w run "waves$w"; done
run base
