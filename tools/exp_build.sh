#!/bin/bash
# tools/exp_build.sh NAME [-D...]: experimental liblutr_NAME.so whose headline tile kernels (10-bit 4:2:0) are compiled with
# extra flags (the LUTR_T2_* knobs of lutr_tile2.hip).  Run with LUTR_LIBRARY=lut_renderer_amd/lib/liblutr_NAME.so.
set -e
NAME=$1; shift
cd "$(dirname "$0")/../lut_renderer_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-honor-nans -fno-slp-vectorize -w --offload-arch=gfx950 -I../../include -I."
mkdir -p build/exp
/opt/rocm/bin/hipcc $FLAGS -DLUTR_T2_WI=1 -DLUTR_T2_WO=1 -DLUTR_T2_X=1 -DLUTR_T2_Y=1 "$@" -c lutr_tile2.hip -o build/exp/$NAME.o
OBJS=$(make -s -f Makefile print-objs | tr " " "\n" | grep -v t2_w11_c11.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/liblutr_$NAME.so $OBJS build/exp/$NAME.o
echo built liblutr_$NAME.so
