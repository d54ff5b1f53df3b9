run() { LUTR_LIBRARY=$1 LUTR_WIN_NODES=$2 LUTR_WAVES_PER_CU=$3 timeout -k 10 120 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/sw.json 2> gpurun_out/sw.err; python -c "
import json;d=json.load(open('gpurun_out/sw.json'));print('$1'.split('/')[-1],'win',$2,'waves/cu',$3,'->',d['value'],'Mpx/s',d['config']['lds_window'])"; }
L=lut_renderer_amd/lib
run $L/liblutr.so 832 12
run $L/liblutr_w4.so 640 16
run $L/liblutr_w4.so 512 16
run $L/liblutr_w5.so 512 20
run $L/liblutr_w5.so 400 20
