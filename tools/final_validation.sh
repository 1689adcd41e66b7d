#!/bin/bash
# One gpurun call on the final build: the whole -m gpu suite, smoke, the bench lines of every workload, the multi-GPU launcher's
# refusal on a 1-GPU box, then the rocprofv3 evidence (tools/collect_profiles.sh).     bash tools/final_validation.sh <tag>
TAG=${1:-r05}; O=gpurun_out/${TAG}_val; mkdir -p $O
if [ -z "${SKIP_TESTS:-}" ]; then
  timeout 2400 python -m pytest ${TESTS:-tests} -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
fi
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash tools/collect_profiles.sh ${TAG}_prof > gpurun_out/${TAG}_prof.log 2>&1
R=$(python -c "import bench; print(bench.PROFILE_ROUND)")
cp gpurun_out/${TAG}_prof/pmc_traffic.json profiles/${R}_pmc_traffic.json 2>/dev/null     # bench.py reads both from profiles/ below
cp gpurun_out/${TAG}_prof/tuning_mnist.txt profiles/${R}_tuning_mnist.txt 2>/dev/null; cp gpurun_out/${TAG}_prof/tuning_celeba.txt profiles/${R}_tuning_celeba.txt 2>/dev/null
cp profiles/${R}_pmc_traffic.json profiles/${R}_tuning_mnist.txt profiles/${R}_tuning_celeba.txt $O/ 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_mnist_driver_cmd.json 2> $O/b.err
python bench.py --workload celeba --steps 5 --warmup 2 > $O/bench_celeba.json 2>> $O/b.err
python bench.py --workload fmnist --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_fmnist.json 2>> $O/b.err
python bench.py --strong --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_strong.json 2>> $O/b.err
python bench.py --batch 50 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_b50.json 2>> $O/b.err
python bench.py --strong --batch 50 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_strong_b50.json 2>> $O/b.err     # the reference's BATCH_SIZE through the coalescing harness
python bench.py --use_bn --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_mnist_use_bn.json 2>> $O/b.err
python bench.py --workload celeba --use_bn --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_celeba_use_bn.json 2>> $O/b.err
python bench.py --gpus 2 --steps 1 --warmup 0 > $O/bench_gpus2.out 2> $O/bench_gpus2.err; echo "bench.py --gpus 2 on this box: exit status $?" | tee $O/bench_gpus2.status; tail -1 $O/bench_gpus2.err
for f in $O/bench_*.json; do echo $f; python -c "import json,sys; d=json.load(open(sys.argv[1])); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['path_frac'], d['roofline']['traffic'], d.get('tuning_id'))" $f; done
python tools/kernel_regs.py > $O/kernel_regs.txt 2>&1
ls gpurun_out/${TAG}_prof | head -40
