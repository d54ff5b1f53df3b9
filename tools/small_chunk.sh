for f in 8 16 32; do for c in 2 4 8; do
LUTR_CHUNK=$c timeout -k 10 100 python bench.py --lean --no-stats --no-other --frames $f --variant vec_lds --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('frames %3d chunk %d %6.1f Gpx/s  %.1f us' % ($f, $c, d['value']/1e3, d['ms_per_step']*1e3))"
done; done
