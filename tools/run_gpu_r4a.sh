#!/bin/bash
# Round 4, call A: the weight-stationary Linear kernels, tuning hand-over, replayed loop graph -- tests first, then A/B benches.
TAG=${1:-r4a}; O=gpurun_out/$TAG; mkdir -p $O
export DG_TUNING_CACHE=$PWD/$O/tuning_cache.txt
T="tests/test_gpu_variants.py::test_latent_turn_kernels_reproduce_the_generic_gemm tests/test_gpu_tuning_graph.py tests/test_gpu_variants.py::test_launch_shape_variants_are_bit_identical tests/test_gpu_mnist.py tests/test_gpu_prepare.py"
DG_TUNING_CACHE= timeout 900 python -m pytest $T -x -q -m gpu > $O/pytest_new.log 2>&1; tail -15 $O/pytest_new.log
summ() { python - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "FAILED", e); continue
    ks = " ".join("%s %.1f" % (k["name"], k["avg_us"]) for k in d["kernels"])
    print("%-40s %8.2f img/s path %.4f | %s" % (f.split("/")[-1], d["value"], d["roofline"]["path_frac"], ks))
PY
}
B="python bench.py --no-cpu-baseline --steps 8 --warmup 3"
for round in 1 2; do
  $B > $O/mnist_default_$round.json 2> $O/mnist_default_$round.err
  $B --opt latent_turn=0 > $O/mnist_lt0_$round.json 2> $O/mnist_lt0_$round.err
done
$B --opt lin_groups_bwd=32 > $O/mnist_bwd32.json 2> $O/err.txt
$B --opt lin_groups_fwd=24 > $O/mnist_fwd24.json 2>> $O/err.txt
$B --opt lin_groups_fwd=8 > $O/mnist_fwd8.json 2>> $O/err.txt
$B --batch 50 > $O/b50_default.json 2>> $O/err.txt
$B --batch 50 --opt graph_max_rows=0 > $O/b50_nograph.json 2>> $O/err.txt
$B --batch 50 --opt graph_max_rows=0 --opt latent_turn=0 > $O/b50_nograph_lt0.json 2>> $O/err.txt
$B --workload celeba --steps 3 --warmup 1 > $O/celeba_default.json 2>> $O/err.txt
$B --workload celeba --steps 3 --warmup 1 --opt latent_turn=0 > $O/celeba_lt0.json 2>> $O/err.txt
$B --opt graph_max_rows=4096 > $O/mnist_graph.json 2>> $O/err.txt
summ $O/*.json | tee $O/summary.txt
tail -3 $O/err.txt
