#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/r03_gputest_e.log 2>&1; echo "pytest rc=$?" | tee -a $O/r03_gputest_e.log
tail -3 $O/r03_gputest_e.log
timeout -k 10 200 python tools/soak.py 45 > $O/r03_soak_e.txt 2>&1; tail -2 $O/r03_soak_e.txt
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/r03_bench_e.json 2> $O/r03_bench_e.err; echo "bench rc=$?"
grep "extra\|other\|tile stats" $O/r03_bench_e.err | cut -c1-230
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_bench_e.json").read().strip().splitlines()[-1])
print("value", d["value"], d["roofline"]["frac"], "other", d["other_precision"]["Mpx_s"], d["other_precision"]["frac"])
PY
