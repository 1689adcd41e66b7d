#!/bin/bash
# One gpurun call: A/B of the kernels against the previous build, then the new tests first, then the rest of the -m gpu suite.
TAG=${1:-g2}; O=gpurun_out/$TAG; mkdir -p $O
bash tools/ab_kernels.sh $TAG tools/ab/lib_r02_kernels.so 2>&1 | tee $O/ab_mnist.txt
NEW="tests/test_gpu_prepare.py tests/test_gpu_cache.py tests/test_gpu_config4.py tests/test_gpu_variants.py tests/test_gpu_parity_tiers.py"
timeout 1200 python -m pytest $NEW -x -q -m gpu -s > $O/pytest_new.log 2>&1; tail -5 $O/pytest_new.log
timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_prepare.py --deselect tests/test_gpu_cache.py --deselect tests/test_gpu_config4.py --deselect tests/test_gpu_variants.py --deselect tests/test_gpu_parity_tiers.py > $O/pytest_rest.log 2>&1; tail -5 $O/pytest_rest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
