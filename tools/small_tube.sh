for f in 8 16 32 64; do for h in 8 5 0; do
LUTR_TUBE_H=$h timeout -k 10 100 python bench.py --lean --no-stats --no-other --frames $f --variant vec_lds --steps 60 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('frames %3d tube_h %d %6.1f Gpx/s  %.1f us' % ($f, $h, d['value']/1e3, d['ms_per_step']*1e3))"
done; done
