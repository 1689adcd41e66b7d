#!/bin/bash
# The 1 / 2 / 4 / 8-GPU sweep BASELINE.json asks for, ready to run on a node that has the GPUs (none was available to the build:
# every ">= 7x at 8 GPUs" statement in this repository is unmeasured on hardware until this script has run).
#   bash tools/scale_sweep.sh [outdir]      -> <outdir>/weak_N.json, strong_N.json (bench.py's line per N), summary.txt
# Refuses -- it never reports an N it did not run on -- when the node shows fewer GPUs than the largest N.
# weak:   every rank projects its own 256-image batch per step (configs[1] per GPU), one all_gather of (idx, best loss)
# strong: one defended evaluation of 10 000 FGSM images sharded over the ranks (configs[4]), one all_gather of (labels, preds, diffs)
# Each line carries `ranks` (size of the RCCL group), ms_per_step_per_rank, tuning_id_per_rank (identical ids = identical job
# lists on every rank, rank 0's choice broadcast), host_cpus_rank0 (the pinned host CPUs) and, from the untimed profiled step,
# the per-layer durations of rank 0.
O=${1:-gpurun_out/scale}; mkdir -p $O
NS=${NS:-"1 2 4 8"}
MAXN=$(for n in $NS; do echo $n; done | sort -n | tail -1)
HAVE=$(python -c "import torch; print(torch.cuda.device_count())")
if [ "$HAVE" -lt "$MAXN" ]; then
  echo "scale_sweep: this node shows $HAVE GPU(s), the sweep needs $MAXN (set NS=\"1 2\" to sweep what is there)" | tee $O/refused.txt
  exit 2
fi
export HSA_ENABLE_IPC_MODE_LEGACY=0
for n in $NS; do
  python bench.py --gpus $n --steps ${STEPS:-10} --warmup ${WARMUP:-3} --no-cpu-baseline > $O/weak_$n.json 2> $O/weak_$n.err || { echo "weak N=$n failed"; tail -3 $O/weak_$n.err; }
  python bench.py --gpus $n --strong --steps 1 --warmup 0 --no-cpu-baseline > $O/strong_$n.json 2> $O/strong_$n.err || { echo "strong N=$n failed"; tail -3 $O/strong_$n.err; }
done
python - $O $NS <<'PY' | tee $O/summary.txt
import json, os, sys
O, ns = sys.argv[1], [int(v) for v in sys.argv[2:]]
for mode in ("weak", "strong"):
    base = None
    for n in ns:
        try:
            d = json.load(open(os.path.join(O, "%s_%d.json" % (mode, n))))
        except Exception as e:
            print("%-6s N=%d: no line (%s)" % (mode, n, e)); continue
        base = base or d["value"] / d["n_gpus"]
        ids = set(d.get("tuning_id_per_rank", []))
        print("%-6s N=%d ranks=%d  %9.1f img/s  x%.2f of N=1 per-GPU  path %.4f  per-rank ms %s  job lists %s  host CPUs rank0 %s" % (
            mode, d["n_gpus"], d.get("ranks", 0), d["value"], d["value"] / base, d["roofline"]["path_frac"],
            d.get("ms_per_step_per_rank"), "identical" if len(ids) == 1 else "DIFFER %s" % sorted(ids), d.get("host_cpus_rank0")))
PY
