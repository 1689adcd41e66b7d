#!/bin/bash
# The -m gpu suite + smoke, as the driver runs them at round end.   bash tools/run_gpu_suite.sh <tag> [pytest targets]
TAG=${1:-suite}; shift; O=gpurun_out/$TAG; mkdir -p $O
T=${@:-tests}
timeout 3000 python -m pytest $T -x -q -m gpu --durations=15 > $O/pytest_gpu.log 2>&1; tail -30 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
