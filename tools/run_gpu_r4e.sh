#!/bin/bash
# Round 4, call E: the rebalanced MNIST tail (mnist_tail_pipe2_kernel) -- bit-identity tests, then A/B against version 1.
TAG=${1:-r4e}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest "tests/test_gpu_variants.py::test_launch_shape_variants_are_bit_identical" -x -q -m gpu > $O/pytest.log 2>&1; tail -6 $O/pytest.log
export DG_TUNING_CACHE=$PWD/$O/tuning_cache.txt
B="python bench.py --no-cpu-baseline --steps 8 --warmup 3"
for round in 1 2; do
  $B > $O/mnist_v2_$round.json 2> $O/err.txt
  $B --opt tail_pipe_version=1 > $O/mnist_v1_$round.json 2>> $O/err.txt
  $B --opt tail_pipe_version=3 > $O/mnist_v3_$round.json 2>> $O/err.txt
done
$B --batch 1250 --steps 3 --warmup 1 --opt tail_pipe_version=3 > $O/b1250_v3.json 2>> $O/err.txt
$B --batch 1250 --steps 3 --warmup 1 > $O/b1250_v2.json 2>> $O/err.txt
$B --batch 1250 --steps 3 --warmup 1 --opt tail_pipe_version=1 > $O/b1250_v1.json 2>> $O/err.txt
python - $O/*.json <<'PY' | tee $O/summary.txt
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "FAILED", e); continue
    ks = " ".join("%s %.1f" % (k["name"], k["avg_us"]) for k in d["kernels"])
    print("%-28s %8.2f img/s path %.4f | %s" % (f.split("/")[-1], d["value"], d["roofline"]["path_frac"], ks))
PY
tail -3 $O/err.txt
