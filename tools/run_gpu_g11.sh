#!/bin/bash
TAG=${1:-g11}; O=gpurun_out/$TAG; mkdir -p $O; rm -f $O/*.json
timeout 600 python -m pytest tests/test_gpu_variants.py tests/test_gpu_mnist.py -x -q -m gpu > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $O/pytest.log
bash tools/ab_kernels.sh $TAG tools/ab/lib_r03d.so 2>&1 | tee $O/ab_mnist.txt
