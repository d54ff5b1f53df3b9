for fmt in yuv422p10le yuv444p10le; do for l in 4 5 6; do for h in 5 0; do
LUTR_LW_LOG2=$l LUTR_TUBE_H=$h timeout -k 10 100 python bench.py --lean --no-stats --fmt $fmt --frames 128 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d.get('other_precision') or {}; print('$fmt lw $l tube $h strict %.0f fast %.0f' % (d['value']/1e3, o.get('Mpx_s',0)/1e3))"
done; done; done
