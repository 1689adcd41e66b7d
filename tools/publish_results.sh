#!/bin/bash
# Copies the summaries of one tools/final_validation.sh run (gpurun_out/<tag>_val, gpurun_out/<tag>_prof) into profiles/<tag>_*.
#   bash tools/publish_results.sh r06
TAG=${1:-r06}; V=gpurun_out/${TAG}_val; P=gpurun_out/${TAG}_prof
for f in $V/bench_*.json; do cp $f profiles/${TAG}_$(basename $f); done
cp $V/bench_gpus2.status profiles/${TAG}_bench_gpus2_refused.txt; tail -2 $V/bench_gpus2.err >> profiles/${TAG}_bench_gpus2_refused.txt
cp $V/kernel_regs.txt profiles/${TAG}_kernel_regs.txt
{ echo "python -m pytest tests -x -q -m gpu on the final build ($(cat $P/build.txt | head -1)):"; grep -v "amdgpu.ids" $V/pytest.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"; tail -1 $V/smoke.log; } > profiles/${TAG}_gpu_suite_final_build.txt
for w in mnist celeba; do
  cp $P/${w}_kernel_stats.csv profiles/${TAG}_${w}_kernel_stats.csv
  cp $P/${w}_kernel_stats_by_layer.csv profiles/${TAG}_${w}_kernel_stats_by_layer.csv
  cp $P/bench_${w}_under_rocprof.json profiles/${TAG}_bench_${w}_under_rocprof.json
  cp $P/tuning_${w}.txt profiles/${TAG}_tuning_${w}.txt
done
cp $P/pmc_traffic.json profiles/${TAG}_pmc_traffic.json
cp $P/pmc_sq.txt profiles/${TAG}_pmc_sq.txt
ls profiles | grep "^${TAG}_" | wc -l
