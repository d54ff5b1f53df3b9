for cfg in "512 16" "640 16" "832 12" "1024 8" "1280 8" "2048 4"; do set -- $cfg; 
LUTR_WIN_NODES=$1 LUTR_WAVES_PER_CU=$2 timeout -k 10 120 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/sw.json 2> gpurun_out/sw.err; 
python -c "
import json;d=json.load(open('gpurun_out/sw.json'));print('win',$1,'waves/cu',$2,'->',d['value'],'Mpx/s',d['config']['lds_window'])"; done
