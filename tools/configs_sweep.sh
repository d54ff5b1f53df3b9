#!/bin/bash
# tools/configs_sweep.sh -- one bench line per BASELINE.json config / format family (1 GPU), as a table.
# Column "other" = the other precision (fast when the line ran strict, the default) timed on the same batch.
run() { timeout -k 10 200 python bench.py --lean --steps 40 --warmup 10 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']; w=c.get('lds_window') or {}; o=d.get('other_precision') or {}
wl=c['workload'].split(',')[0]
chain=('pc->tv prologue + ' if c.get('range_src')=='pc' else '')+'lut3d'
print('| %-32s | %2d^3 | %-11s | %-8s | %4d | %-22s | %-6s | %9.0f | %6.0f | %5.1f%% | %9s | %s | %s |' % (wl, c['lut_size'], c['interp'], c['distribution'], c['frames_per_gpu'], chain, c['precision'], d['value'], r['achieved'], 100*r['frac'], ('%.0f' % o['Mpx_s']) if o else '-', c['kernel'], '%d/%d/%d/%d/%d/%d' % (w.get('tiles',0), w.get('tube_tiles',0), w.get('mixed_tiles',0), w.get('level2_tiles',0), w.get('misses',0), w.get('global_tiles',0))))"; }
echo "| frames | LUT | interp | content | frames/step | chain | precision | Mpx/s | GB/s | of 8 TB/s | other precision Mpx/s | kernel | tiles/tube/mixed/level-2/restage attempts/gather |"
echo "|---|---|---|---|---|---|---|---|---|---|---|---|---|"
run --size uhd --fmt yuv420p10le --interp tetrahedral
run --size uhd --fmt yuv420p10le --interp trilinear
run --size uhd --fmt yuv420p10le --interp tetrahedral --dist noise8 --frames 64
run --size uhd --fmt yuv420p10le --interp tetrahedral --dist noise16 --frames 64
run --size uhd --fmt yuv420p10le --interp tetrahedral --dist vivid --frames 64
run --size uhd --fmt yuv420p10le --interp tetrahedral --dist uniform --frames 64
run --size uhd --fmt yuv420p10le --interp tetrahedral --lut 65
run --size uhd --fmt yuv420p10le --interp tetrahedral --lut 17
run --size uhd --fmt yuv420p10le --interp tetrahedral --lut 17 --dist uniform --frames 64
run --size uhd --fmt yuv420p10le --out-fmt yuv420p --interp tetrahedral
run --size uhd --fmt yuv420p10le --range-src pc --interp tetrahedral
run --size uhd --fmt yuv420p10le --range-src pc --out-fmt yuv420p --interp tetrahedral
run --size 1080p --fmt yuv420p --interp trilinear --frames 512
run --size 1080p --fmt yuv420p --interp tetrahedral --frames 512
run --size 1080p --fmt yuv420p --interp tetrahedral --frames 64
run --size 8k --fmt yuv420p10le --interp tetrahedral --frames 64
run --size 8k --fmt yuv420p10le --interp trilinear --frames 64
run --size uhd --fmt yuv422p10le --interp tetrahedral
run --size uhd --fmt yuv444p10le --interp tetrahedral --frames 128
run --size uhd --fmt gbrp10le --interp tetrahedral --frames 128
run --size uhd --fmt gbrp10le --interp tetrahedral --frames 16
run --size uhd --fmt gbrp10le --interp trilinear --frames 128
run --size uhd --fmt gbrp --interp tetrahedral --frames 128
run --size uhd --fmt gbrp10le --interp nearest --frames 128
run --size uhd --fmt rgb24 --interp tetrahedral --frames 128
run --size uhd --fmt rgba --interp tetrahedral --frames 128
run --size uhd --fmt rgb48le --interp tetrahedral --frames 64
run --size uhd --fmt rgba64le --interp tetrahedral --frames 64
run --size uhd --fmt rgb24 --interp trilinear --frames 128
run --size uhd --fmt bgra --interp tetrahedral --frames 128
run --size uhd --fmt gbrp16le --interp tetrahedral --frames 64
