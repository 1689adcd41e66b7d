#!/bin/bash
TAG=${1:-g6}; O=gpurun_out/$TAG; mkdir -p $O
bash tools/ab_kernels.sh $TAG tools/ab/lib_r03b.so --use_bn 2>&1 | tee $O/ab_use_bn.txt
timeout 600 python -m pytest tests/test_gpu_celeba_bn.py tests/test_gpu_mnist.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
