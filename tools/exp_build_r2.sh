#!/bin/bash
# tools/exp_build_r2.sh NAME [-D...]: experimental liblutr_NAME.so whose RGB tube kernels (lutr_rgb2.hip, every layout) are compiled
# with extra flags (the LUTR_R2_* knobs).  Run with LUTR_LIBRARY=lut_renderer_amd/lib/liblutr_NAME.so.
set -e
NAME=$1; shift
cd "$(dirname "$0")/../lut_renderer_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-honor-nans -fno-slp-vectorize -w --offload-arch=gfx950 -I../../include -I."
mkdir -p build/exp
R2=""
for l in 0 1 2 3 4 5 6 7; do
  /opt/rocm/bin/hipcc $FLAGS -DLUTR_R2_LAYOUT=$l "$@" -c lutr_rgb2.hip -o build/exp/${NAME}_ly$l.o &
  R2="$R2 build/exp/${NAME}_ly$l.o"
done
wait
OBJS=$(make -s -f Makefile print-objs | tr " " "\n" | grep -v r2_ly)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/liblutr_$NAME.so $OBJS $R2
echo built liblutr_$NAME.so
