#!/bin/bash
TAG=${1:-g8}; O=gpurun_out/$TAG; mkdir -p $O; rm -f $O/celeba_*.json
C="--workload celeba --steps 3 --warmup 1 --no-cpu-baseline"
for r in 1 2; do
  for W in 512 2228736 4325888 6423040 8520192; do
    timeout 300 python bench.py $C --opt tail_fwd_split=$W > $O/celeba_s${W}_$r.json 2> $O/celeba_s${W}_$r.err
  done
done
python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(os.path.join(sys.argv[1],"celeba_*.json"))):
    try:
        d=json.load(open(f)); ks=" ".join("%s %.1f"%(k["name"],k["avg_us"]) for k in d["kernels"] if k["name"] in ("T6f","T6b"))
        print("%-26s %8.2f img/s path %.4f | %s" % (os.path.basename(f), d["value"], d["roofline"]["path_frac"], ks))
    except Exception as e: print(f,"FAILED",e)
PY
