#!/bin/bash
O=gpurun_out; mkdir -p $O
bash tools/profile.sh uhd420p10_tetra_natural_f256_strict > $O/prof_a.log 2>&1; echo "strict done"
bash tools/profile.sh uhd420p10_tetra_natural_f256_fast --precision fast > $O/prof_b.log 2>&1; echo "fast done"
bash tools/profile.sh uhd_rgb24_tetra_natural_f128 --fmt rgb24 --frames 128 > $O/prof_c.log 2>&1; echo "rgb24 done"
bash tools/profile.sh uhd_gbrp10_tetra_natural_f128 --fmt gbrp10le --frames 128 > $O/prof_d.log 2>&1; echo "gbrp10 done"
