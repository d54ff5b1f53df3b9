#!/bin/bash
# tools/kernel_usage.sh [-D...]: registers / spills / scratch of the headline tile kernels (10-bit 4:2:0) as the compiler reports them
# (-Rpass-analysis=kernel-resource-usage), for a set of LUTR_T2_* flags.  CPU only (hipcc cross-compiles).
cd "$(dirname "$0")/../lut_renderer_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-honor-nans -fno-slp-vectorize -w --offload-arch=gfx950 -I../../include -I."
mkdir -p build/exp
/opt/rocm/bin/hipcc $FLAGS -DLUTR_T2_WI=${WI:-1} -DLUTR_T2_WO=${WO:-1} -DLUTR_T2_X=${CX:-1} -DLUTR_T2_Y=${CY:-1} "$@" -S --cuda-device-only -Rpass-analysis=kernel-resource-usage \
    lutr_tile2.hip -o build/exp/usage.s 2>&1 | python3 -c "
import re,sys
txt=sys.stdin.read()
for blk in txt.split('Function Name: ')[1:]:
    name=blk.split('\n')[0]
    m=re.search(r'k_yuv_tile2I((?:Li\d+E)+)', name)
    if m: m=re.match(r'(.*)', '<'+','.join(re.findall(r'Li(\d+)E', m.group(1)))+'>  [win,wout,csx,csy,interp,pre,variant]')
    g=lambda k: (re.search(k+r': (\d+)', blk) or [0,'?'])[1]
    print('%-28s VGPRs %s  AGPRs %s  SGPRs %s  sgpr-spill %s  vgpr-spill %s  scratch %s  occupancy %s  LDS %s' % (m.group(1) if m else name[:28], g('VGPRs'), g('AGPRs'), g('SGPRs'), g('SGPRs Spill'), g('VGPRs Spill'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))
"
