#!/bin/bash
TAG=${1:-r4c}; O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_variants.py::test_latent_turn_kernels_reproduce_the_generic_gemm "tests/test_gpu_variants.py::test_launch_shape_variants_are_bit_identical" tests/test_gpu_tuning_graph.py -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for op in F1 B1; do
  python tools/lin_trace.py $op > $O/lin_trace_${op}_2560.txt 2>&1
  python tools/lin_trace.py $op B=50 > $O/lin_trace_${op}_500.txt 2>&1
done
python tools/lin_trace.py B1 lin_groups_bwd=32 > $O/lin_trace_B1_2560_g32.txt 2>&1
tail -n 9 $O/lin_trace_*.txt | grep -v amdgpu.ids
export DG_TUNING_CACHE=$PWD/$O/tuning_cache.txt
B="python bench.py --no-cpu-baseline --steps 8 --warmup 3"
for round in 1 2; do
  $B > $O/mnist_default_$round.json 2> $O/err.txt
  $B --opt latent_turn=0 > $O/mnist_lt0_$round.json 2>> $O/err.txt
done
$B --opt lin_groups_bwd=32 > $O/mnist_bwd32.json 2>> $O/err.txt
$B --batch 50 > $O/b50_default.json 2>> $O/err.txt
$B --batch 50 --opt latent_turn=0 > $O/b50_lt0.json 2>> $O/err.txt
$B --workload celeba --steps 3 --warmup 1 > $O/celeba_default.json 2>> $O/err.txt
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
