#!/bin/bash
# Short CelebA counterpart of collect_mnist_quick.sh: kernel stats of the CelebA bench command and the FETCH / WRITE passes.
set -u
TAG=${1:-profq}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BUILD=$(python -c "import bench; print(bench.build_id())")
echo "build $BUILD" > $OUT/build_celeba.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o celeba -- python bench.py --workload celeba --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_celeba_under_rocprof.json 2> $OUT/stats_celeba.err
find $OUT/stats -name "celeba_kernel_stats.csv" -exec cp {} $OUT/ \;
python -c "import json; d=json.load(open('$OUT/bench_celeba_under_rocprof.json')); print(d['value'], d['roofline']['frac'], d['roofline']['path_frac'], [(k['name'],k['avg_us']) for k in d['kernels']])"
CMD="python bench.py --workload celeba --steps 1 --warmup 1 --rec_iters 4 --no-cpu-baseline --no-profile"
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc -o celeba_fetch -- $CMD > /dev/null 2> $OUT/pmc_celeba_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc -o celeba_write -- $CMD > /dev/null 2> $OUT/pmc_celeba_write.err
F=$(find $OUT/pmc -name "celeba_fetch*results.db" | head -1); Wd=$(find $OUT/pmc -name "celeba_write*results.db" | head -1)
python tools/pmc_traffic.py celeba $F $Wd $BUILD > $OUT/pmc_traffic_celeba.json 2> $OUT/pmc_traffic_celeba.err
find $OUT -name "*.db" -delete
