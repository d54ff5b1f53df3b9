#!/bin/bash
# round-3 evidence: rocprofv3 passes for the headline (strict and fast), planar RGB and packed RGB; default bench line
O=gpurun_out; mkdir -p $O
bash tools/profile.sh uhd420p10_tetra_natural_f256_strict > $O/prof_a.log 2>&1; echo "strict done"
bash tools/profile.sh uhd420p10_tetra_natural_f256_fast --precision fast > $O/prof_b.log 2>&1; echo "fast done"
bash tools/profile.sh uhd_rgb24_tetra_natural_f128 --fmt rgb24 --frames 128 > $O/prof_c.log 2>&1; echo "rgb24 done"
bash tools/profile.sh uhd_gbrp10_tetra_natural_f128 --fmt gbrp10le --frames 128 > $O/prof_d.log 2>&1; echo "gbrp10 done"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/r03_bench_default.json 2> $O/r03_bench_default.err; echo "bench rc=$?"
tail -c 400 $O/r03_bench_default.json
