#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r03_gputest_i.log 2>&1; echo "pytest rc=$?" | tee -a $O/r03_gputest_i.log
tail -30 $O/r03_gputest_i.log | cut -c1-250
