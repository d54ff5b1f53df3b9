for f in 4 8 16 32 64 128; do for v in vec_global vec_lds; do
timeout -k 10 100 python bench.py --lean --no-stats --no-other --fmt gbrp10le --frames $f --variant $v --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('gbrp10le frames %3d %-10s %6.1f Gpx/s %5.0f GB/s %s' % ($f, '$v', d['value']/1e3, d['roofline']['achieved'], d['config']['kernel']))"
done; done
