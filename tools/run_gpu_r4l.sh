#!/bin/bash
# Round 4, call L: tapered job lists among the timed candidates (dg_plan.h JobModel::taper): tuner table + A/B.
TAG=${1:-r4l}; O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest "tests/test_gpu_variants.py::test_launch_shape_variants_are_bit_identical" tests/test_gpu_tuning_graph.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
B="python bench.py --no-cpu-baseline --steps 8 --warmup 3"
DG_TUNE_VERBOSE=1 $B > $O/mnist_taper_verbose.json 2> $O/tune_table.txt
grep "kept" $O/tune_table.txt
for round in 1 2 3; do
  DG_TUNING_CACHE=$PWD/$O/cache_taper.txt $B > $O/mnist_taper_$round.json 2> $O/err.txt
  DG_TUNING_CACHE=$PWD/$O/cache_plain.txt $B --opt jobs.taper_tune=0 > $O/mnist_plain_$round.json 2>> $O/err.txt
done
DG_TUNING_CACHE=$PWD/$O/cache_celeba_taper.txt $B --workload celeba --steps 3 --warmup 1 > $O/celeba_taper.json 2>> $O/err.txt
DG_TUNING_CACHE=$PWD/$O/cache_celeba_plain.txt $B --workload celeba --steps 3 --warmup 1 --opt jobs.taper_tune=0 > $O/celeba_plain.json 2>> $O/err.txt
$B --batch 50 > $O/b50_taper.json 2>> $O/err.txt
$B --batch 50 --opt jobs.taper_tune=0 > $O/b50_plain.json 2>> $O/err.txt
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
