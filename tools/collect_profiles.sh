#!/bin/bash
# Run on the GPU box (through gpurun): the rocprofv3 evidence bench.py's roofline line is checked against.
#   bash tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>/...
# --kernel-trace --stats of the driver's bench command; FETCH_SIZE / WRITE_SIZE / SQ counters in SEPARATE --pmc passes
# (no tracing flags combined with --pmc).  Every pass of a workload shares ONE tuning cache (DG_TUNING_CACHE): the first run
# times the job lists, the others install its choices, so the bench line, the kernel trace and the PMC bytes describe the same
# lists (bench.py's "tuning_id"; tools/pmc_traffic.py records it next to the build id).
set -u
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BUILD=$(python -c "import bench; print(bench.build_id())")
echo "build $BUILD" > $OUT/build.txt
tid() { python -c "import sys; from defensegan_amd.gan import tuning_text_id; print(tuning_text_id(open(sys.argv[1]).read()))" $1; }
export DG_TUNING_CACHE=$PWD/$OUT/tuning_mnist.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_mnist_driver_cmd.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o mnist -- python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_mnist_under_rocprof.json 2> $OUT/stats_mnist.err
export DG_TUNING_CACHE=$PWD/$OUT/tuning_celeba.txt
python bench.py --workload celeba --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_celeba.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o celeba -- python bench.py --workload celeba --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_celeba_under_rocprof.json 2> $OUT/stats_celeba.err
for W in mnist celeba; do
  export DG_TUNING_CACHE=$PWD/$OUT/tuning_$W.txt
  CMD="python bench.py --workload $W --steps 1 --warmup 1 --rec_iters 4 --no-cpu-baseline --no-profile"
  rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc -o ${W}_fetch -- $CMD > /dev/null 2> $OUT/pmc_${W}_fetch.err
  rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc -o ${W}_write -- $CMD > /dev/null 2> $OUT/pmc_${W}_write.err
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM GRBM_GUI_ACTIVE -d $OUT/pmc -o ${W}_sq -- $CMD > /dev/null 2> $OUT/pmc_${W}_sq.err
  F=$(find $OUT/pmc -name "${W}_fetch*results.db" | head -1); Wd=$(find $OUT/pmc -name "${W}_write*results.db" | head -1)
  python tools/pmc_traffic.py $W $F $Wd $BUILD $(tid $OUT/tuning_$W.txt) > $OUT/pmc_traffic_$W.json 2>> $OUT/pmc_traffic.err
done
unset DG_TUNING_CACHE
python - $OUT <<'PY'
import json, sys
O = sys.argv[1]
doc = {"_comment": "HBM-side bytes per launch, by LAYER (position in the fixed launch order of a GD iteration) and by kernel symbol: "
       "(2 x FETCH_SIZE + WRITE_SIZE) KB x 1024, separate rocprofv3 --pmc passes, dispatches of the last projection call only "
       "(tools/collect_profiles.sh, tools/pmc_traffic.py).  bench.py quotes roofline.traffic from by_layer only when its build id AND "
       "its tuning id (which job lists ran) equal the ones below.", "builds": {}, "tuning_ids": {}}
for w in ("mnist", "celeba"):
    try:
        d = json.load(open("%s/pmc_traffic_%s.json" % (O, w)))
    except Exception as e:
        print("no traffic for", w, e); continue
    doc[w] = d[w]; doc["builds"].update(d.get("builds", {})); doc["tuning_ids"].update(d.get("tuning_ids", {}))
json.dump(doc, open(O + "/pmc_traffic.json", "w"), indent=1)
PY
python tools/pmc_summary.py $(find $OUT/pmc -name "*_sq*results.db") > $OUT/pmc_sq.txt 2> $OUT/pmc_sq.err
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/ \;
# per-LAYER rows (one symbol serves several layers; candidate-list launches folded out): tools/kernel_trace_by_layer.py
for W in mnist celeba; do
  T=$(find $OUT/stats -name "${W}_kernel_trace.csv" | head -1)
  [ -n "$T" ] && python tools/kernel_trace_by_layer.py $T $OUT/bench_${W}_under_rocprof.json > $OUT/${W}_kernel_stats_by_layer.csv 2>> $OUT/by_layer.err
done
find $OUT -name "*.db" -delete
rm -rf $OUT/stats/*/*_agent_info.csv
ls -la $OUT
