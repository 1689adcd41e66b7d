#!/bin/bash
TAG=${1:-g7}; O=gpurun_out/$TAG; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_variants.py -x -q -m gpu -k "role_split" > $O/pytest_split.log 2>&1; echo "split test rc=$?"; tail -5 $O/pytest_split.log
if grep -q "passed" $O/pytest_split.log && ! grep -q "failed" $O/pytest_split.log; then
  C="--workload celeba --steps 3 --warmup 1 --no-cpu-baseline"
  for r in 1 2; do
    timeout 300 python bench.py $C > $O/celeba_band_$r.json 2> $O/celeba_band_$r.err
    for W in 512 256 768; do
      timeout 300 python bench.py $C --opt tail_fwd_split=$W > $O/celeba_split${W}_$r.json 2> $O/celeba_split${W}_$r.err
    done
  done
  python - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(os.path.join(sys.argv[1],"celeba_*.json"))):
    try:
        d=json.load(open(f)); ks=" ".join("%s %.1f"%(k["name"],k["avg_us"]) for k in d["kernels"] if k["name"] in ("T6f","T6b","F5","B5"))
        print("%-26s %8.2f img/s path %.4f | %s" % (os.path.basename(f), d["value"], d["roofline"]["path_frac"], ks))
    except Exception as e: print(f,"FAILED",e)
PY
fi
