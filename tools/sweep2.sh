run() { LUTR_LIBRARY=$1 timeout -k 10 120 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/sw.json 2> gpurun_out/sw.err; python -c "
import json;d=json.load(open('gpurun_out/sw.json'));print('$1'.split('/')[-1],'->',d['value'],'Mpx/s')"; }
L=lut_renderer_amd/lib
run $L/liblutr.so
run $L/liblutr_a1.so
run $L/liblutr_a2.so
run $L/liblutr_a4.so
run $L/liblutr_a7.so
