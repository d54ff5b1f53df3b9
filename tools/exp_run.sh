#!/bin/bash
# tools/exp_run.sh NAME...: headline bench (strict = value, fast = other_precision) under each experimental library (tools/exp_build.sh)
for n in "$@"; do
  lib=lut_renderer_amd/lib/liblutr_$n.so; [ "$n" = base ] && lib=lut_renderer_amd/lib/liblutr.so
  LUTR_LIBRARY=$lib timeout -k 10 120 python bench.py --lean --no-stats 2>/dev/null | tail -1 > gpurun_out/exp_$n.json
  python - <<PY
import json
d=json.load(open("gpurun_out/exp_$n.json")); o=d.get("other_precision") or {}
print("%-10s strict %.1f  fast %.1f" % ("$n", d["value"]/1e3, o.get("Mpx_s",0)/1e3))
PY
done
