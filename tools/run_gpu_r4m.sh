#!/bin/bash
# Round 4, call M: the momentum update folded into the Linear backward launch (option update_fold; dg_linear.hip): bit-identity
# over many steps, then A/B on one box with the same installed job lists (profiles/r04_tuning_*.txt).
TAG=${1:-r4m}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest "tests/test_gpu_variants.py::test_folded_update_is_bit_identical_over_many_steps" \
    "tests/test_gpu_variants.py::test_launch_shape_variants_are_bit_identical" -x -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
B="python bench.py --no-cpu-baseline --steps 8 --warmup 3"
for round in 1 2 3; do
  $B > $O/mnist_plain_$round.json 2> $O/err.txt
  $B --opt update_fold=1 > $O/mnist_fold_$round.json 2>> $O/err.txt
done
$B --workload celeba --steps 3 --warmup 1 > $O/celeba_plain.json 2>> $O/err.txt
$B --workload celeba --steps 3 --warmup 1 --opt update_fold=1 > $O/celeba_fold.json 2>> $O/err.txt
$B --batch 50 > $O/b50_plain.json 2>> $O/err.txt
$B --batch 50 --opt update_fold=1 > $O/b50_fold.json 2>> $O/err.txt
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
