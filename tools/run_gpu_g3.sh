#!/bin/bash
TAG=${1:-g3}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
bash tools/ab_kernels.sh $TAG tools/ab/lib_r03a.so --workload celeba 2>&1 | tee $O/ab_celeba.txt
timeout 1500 python -m pytest tests/test_gpu_parity_tiers.py tests/test_gpu_celeba_bn.py tests/test_gpu_variants.py tests/test_gpu_fullsize.py -x -q -m gpu -s > $O/pytest.log 2>&1; tail -5 $O/pytest.log
rocprofv3 -L > $O/counters.txt 2>&1
CMD="python bench.py --steps 1 --warmup 1 --rec_iters 4 --no-cpu-baseline --no-profile"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc -o mnist_stall -- $CMD > /dev/null 2> $O/pmc_stall.err
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc -o mnist_inst -- $CMD > /dev/null 2> $O/pmc_inst.err
python tools/pmc_summary.py $(find $O/pmc -name "*results.db") > $O/pmc_stall.txt 2>> $O/pmc_stall.err
find $O -name "*.db" -delete
tail -30 $O/pmc_stall.txt
