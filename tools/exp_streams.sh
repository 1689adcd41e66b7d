#!/bin/bash
# Experiment (one gpurun call = one box): row groups on several streams x job-list policy.  bash tools/exp_streams.sh <tag>
TAG=${1:-exp}; O=gpurun_out/$TAG; mkdir -p $O
run() { # name, bench args...
  local name=$1; shift
  python bench.py --no-cpu-baseline --no-profile "$@" > $O/$name.json 2> $O/$name.err
  python - $O/$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("%-34s %9.2f img/s  %8.3f ms/step" % (sys.argv[2], d["value"], d["ms_per_step"]))
except Exception as e:
    print("%-34s FAILED %s" % (sys.argv[2], e))
PY
}
WH="--opt jobs.slack=1e30 --opt jobs.min_level=0"
M="--steps 8 --warmup 2"
run mnist_base $M
run mnist_s2 $M --opt two_streams=2
run mnist_s2_whole $M --opt two_streams=2 $WH
run mnist_s3_whole $M --opt two_streams=3 $WH
run mnist_s4 $M --opt two_streams=4
run mnist_s4_whole $M --opt two_streams=4 $WH
run mnist_s4_l1 $M --opt two_streams=4 --opt jobs.slack=1e30 --opt jobs.min_level=1
run mnist_s6_whole $M --opt two_streams=6 $WH
run mnist_s8_whole $M --opt two_streams=8 $WH
run mnist_base2 $M
C="--workload celeba --steps 3 --warmup 1"
run celeba_base $C
run celeba_s2 $C --opt two_streams=2
run celeba_s2_whole $C --opt two_streams=2 $WH
run celeba_s4_whole $C --opt two_streams=4 $WH
run celeba_s8_whole $C --opt two_streams=8 $WH
B="--batch 50 --steps 20 --warmup 3"
run b50_base $B
run b50_s2_whole $B --opt two_streams=2 --opt two_stream_min_rows=64 $WH
run b50_s2_l2 $B --opt two_streams=2 --opt two_stream_min_rows=64 --opt jobs.slack=1e30 --opt jobs.min_level=2
run b50_s5_l2 $B --opt two_streams=5 --opt two_stream_min_rows=64 --opt jobs.slack=1e30 --opt jobs.min_level=2
