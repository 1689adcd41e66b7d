#!/bin/bash
# Run on the GPU box (through gpurun): the rocprofv3 evidence bench.py's roofline line is checked against.
#   bash tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>/...
# --kernel-trace --stats of the driver's bench command; FETCH_SIZE / WRITE_SIZE / SQ counters in SEPARATE --pmc passes
# (no tracing flags combined with --pmc).
set -u
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BUILD=$(python -c "import bench; print(bench.build_id())")
echo "build $BUILD" > $OUT/build.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o mnist -- python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_mnist_under_rocprof.json 2> $OUT/stats_mnist.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o celeba -- python bench.py --workload celeba --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_celeba_under_rocprof.json 2> $OUT/stats_celeba.err
# the USE_BN: True variant (not a BASELINE config): bench line + kernel stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o mnist_use_bn -- python bench.py --use_bn --steps 5 --warmup 2 > $OUT/bench_mnist_use_bn_under_rocprof.json 2> $OUT/stats_mnist_use_bn.err
python bench.py --workload celeba --use_bn --steps 3 --warmup 1 > $OUT/bench_celeba_use_bn.json 2>> $OUT/stats_mnist_use_bn.err
for W in mnist celeba; do
  CMD="python bench.py --workload $W --steps 1 --warmup 1 --rec_iters 4 --no-cpu-baseline --no-profile"
  rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc -o ${W}_fetch -- $CMD > /dev/null 2> $OUT/pmc_${W}_fetch.err
  rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc -o ${W}_write -- $CMD > /dev/null 2> $OUT/pmc_${W}_write.err
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM GRBM_GUI_ACTIVE -d $OUT/pmc -o ${W}_sq -- $CMD > /dev/null 2> $OUT/pmc_${W}_sq.err
done
find $OUT -name "*.db" | head -20 > $OUT/dbs.txt
F=$(find $OUT/pmc -name "mnist_fetch*results.db" | head -1); Wd=$(find $OUT/pmc -name "mnist_write*results.db" | head -1)
python tools/pmc_traffic.py mnist $F $Wd $BUILD > $OUT/pmc_traffic_mnist.json 2> $OUT/pmc_traffic.err
F=$(find $OUT/pmc -name "celeba_fetch*results.db" | head -1); Wd=$(find $OUT/pmc -name "celeba_write*results.db" | head -1)
python tools/pmc_traffic.py celeba $F $Wd $BUILD > $OUT/pmc_traffic_celeba.json 2>> $OUT/pmc_traffic.err
python tools/pmc_summary.py $(find $OUT/pmc -name "*_sq*results.db") > $OUT/pmc_sq.txt 2> $OUT/pmc_sq.err
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/ \;
# per-LAYER rows (one symbol serves several layers; candidate-list launches folded out): tools/kernel_trace_by_layer.py
for W in mnist celeba mnist_use_bn; do
  T=$(find $OUT/stats -name "${W}_kernel_trace.csv" | head -1)
  [ -n "$T" ] && python tools/kernel_trace_by_layer.py $T $OUT/bench_${W}_under_rocprof.json > $OUT/${W}_kernel_stats_by_layer.csv 2>> $OUT/by_layer.err
done
find $OUT -name "*.db" -delete
ls -la $OUT
