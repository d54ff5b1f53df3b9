#!/bin/bash
# tools/exp_run.sh NAME...: headline bench (fast + strict) under each experimental library (tools/exp_build.sh)
for n in "$@"; do
  lib=lut_renderer_amd/lib/liblutr_$n.so; [ "$n" = base ] && lib=lut_renderer_amd/lib/liblutr.so
  LUTR_LIBRARY=$lib timeout -k 10 120 python bench.py --no-cpu-baseline --no-extra --no-strong --no-stats 2>/dev/null | tail -1 > gpurun_out/exp_$n.json
  python - <<PY
import json
d=json.load(open("gpurun_out/exp_$n.json")); o=d.get("other_precision") or {}
print("%-8s fast %.1f  strict %.1f" % ("$n", d["value"]/1e3, o.get("Mpx_s",0)/1e3))
PY
done
