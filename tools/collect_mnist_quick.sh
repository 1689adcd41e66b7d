#!/bin/bash
# Short form of collect_profiles.sh for a small GPU budget: the MNIST driver command under rocprofv3 --stats and the MNIST
# FETCH_SIZE / WRITE_SIZE / SQ passes only.     bash tools/collect_mnist_quick.sh <tag>  -> gpurun_out/<tag>/
set -u
TAG=${1:-profq}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BUILD=$(python -c "import bench; print(bench.build_id())")
echo "build $BUILD" > $OUT/build.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o mnist -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_mnist_under_rocprof.json 2> $OUT/stats_mnist.err
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/ \;
python -c "import json; d=json.load(open('$OUT/bench_mnist_under_rocprof.json')); print(d['value'], d['roofline']['frac'], d['roofline']['path_frac'], [(k['name'],k['avg_us']) for k in d['kernels']])"
CMD="python bench.py --workload mnist --steps 1 --warmup 1 --rec_iters 4 --no-cpu-baseline --no-profile"
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc -o mnist_fetch -- $CMD > /dev/null 2> $OUT/pmc_mnist_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc -o mnist_write -- $CMD > /dev/null 2> $OUT/pmc_mnist_write.err
F=$(find $OUT/pmc -name "mnist_fetch*results.db" | head -1); Wd=$(find $OUT/pmc -name "mnist_write*results.db" | head -1)
python tools/pmc_traffic.py mnist $F $Wd $BUILD > $OUT/pmc_traffic_mnist.json 2> $OUT/pmc_traffic.err
find $OUT/pmc -name "*.db" -delete
timeout 100 python -m pytest tests/test_gpu_variants.py -x -q -m gpu > $OUT/pytest_variants.log 2>&1; tail -1 $OUT/pytest_variants.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM GRBM_GUI_ACTIVE -d $OUT/pmc -o mnist_sq -- $CMD > /dev/null 2> $OUT/pmc_mnist_sq.err
python tools/pmc_summary.py $(find $OUT/pmc -name "*_sq*results.db") > $OUT/pmc_sq.txt 2> $OUT/pmc_sq.err
find $OUT -name "*.db" -delete
