#!/bin/bash
TAG=${1:-r4b}; O=gpurun_out/$TAG; mkdir -p $O
for op in F1 B1; do
  python tools/lin_trace.py $op > $O/lin_trace_${op}_2560.txt 2>&1
  python tools/lin_trace.py $op B=50 > $O/lin_trace_${op}_500.txt 2>&1
done
python tools/lin_trace.py B1 lin_groups_bwd=32 > $O/lin_trace_B1_2560_g32.txt 2>&1
python tools/lin_trace.py F1 lin_groups_fwd=8 > $O/lin_trace_F1_2560_g8.txt 2>&1
tail -n 12 $O/lin_trace_*.txt
