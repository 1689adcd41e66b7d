#!/bin/bash
# A/B on ONE box: the product library of this tree against another build of it (tools/ab/<name>.so, same C ABI), alternating.
#   bash tools/ab_kernels.sh <tag> <other.so> [bench args]      -> gpurun_out/<tag>/ab_*.json + a per-layer table
TAG=${1:-ab}; OTHER=${2:-tools/ab/lib_r02_kernels.so}; shift 2
O=gpurun_out/$TAG; mkdir -p $O
LIB=defensegan_amd/lib/libdefensegan_hip.so
cp $LIB /tmp/dg_new.so
for round in 1 2; do
  for v in new other; do
    if [ $v = new ]; then cp /tmp/dg_new.so $LIB; else cp $OTHER $LIB; fi
    python bench.py --no-cpu-baseline --steps 8 --warmup 2 "$@" > $O/ab_${v}_$round.json 2> $O/ab_${v}_$round.err
  done
done
cp /tmp/dg_new.so $LIB
python - $O <<'PY'
import json, sys, glob, os
O = sys.argv[1]
for f in sorted(glob.glob(os.path.join(O, "ab_*.json"))):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e); continue
    ks = " ".join("%s %.1f" % (k["name"], k["avg_us"]) for k in d["kernels"])
    print("%-18s %8.2f img/s  path %.4f | %s" % (os.path.basename(f), d["value"], d["roofline"]["path_frac"], ks))
PY
