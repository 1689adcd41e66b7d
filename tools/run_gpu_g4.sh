#!/bin/bash
# XCD-locality job order: what the timing keeps, A/B against the order switched off, fabric traffic and clocks (CelebA + MNIST)
TAG=${1:-g4}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
show() { python - "$@" <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.load(open(f)); ks=" ".join("%s %.1f"%(k["name"],k["avg_us"]) for k in d["kernels"])
        print("%-28s %8.2f img/s path %.4f | %s" % (f.split("/")[-1], d["value"], d["roofline"]["path_frac"], ks))
    except Exception as e: print(f, "FAILED", e)
PY
}
C="--workload celeba --steps 3 --warmup 1 --no-cpu-baseline"
for r in 1 2; do
  DG_TUNE_VERBOSE=1 python bench.py $C > $O/celeba_xcd_$r.json 2> $O/celeba_xcd_$r.tune
  python bench.py $C --opt jobs.xcd_head=0 > $O/celeba_off_$r.json 2> $O/celeba_off_$r.err
done
show $O/celeba_xcd_1.json $O/celeba_off_1.json $O/celeba_xcd_2.json $O/celeba_off_2.json | tee $O/summary.txt
grep "kept" $O/celeba_xcd_1.tune | tee -a $O/summary.txt
DG_TUNE_VERBOSE=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/mnist_xcd.json 2> $O/mnist_xcd.tune
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --opt jobs.xcd_head=0 > $O/mnist_off.json 2> $O/mnist_off.err
show $O/mnist_xcd.json $O/mnist_off.json | tee -a $O/summary.txt
grep "kept" $O/mnist_xcd.tune | tee -a $O/summary.txt
CMD="python bench.py --workload celeba --steps 1 --warmup 1 --rec_iters 4 --no-cpu-baseline --no-profile"
for V in xcd off; do
  OPT=""; [ $V = off ] && OPT="--opt jobs.xcd_head=0"
  rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/pmc -o celeba_fetch_$V -- $CMD $OPT > /dev/null 2> $O/pmc_fetch_$V.err
  rocprofv3 --pmc WRITE_SIZE -d $O/pmc -o celeba_write_$V -- $CMD $OPT > /dev/null 2> $O/pmc_write_$V.err
  F=$(find $O/pmc -name "celeba_fetch_${V}*results.db" | head -1); W=$(find $O/pmc -name "celeba_write_${V}*results.db" | head -1)
  python tools/pmc_traffic.py celeba $F $W > $O/pmc_traffic_celeba_$V.json 2>> $O/pmc.err
  python tools/pmc_summary.py $F > $O/pmc_fetch_$V.txt 2>> $O/pmc.err
done
find $O -name "*.db" -delete
python - $O <<'PY' | tee -a $O/summary.txt
import json,sys
O=sys.argv[1]
a=json.load(open(O+"/pmc_traffic_celeba_xcd.json"))["celeba"]; b=json.load(open(O+"/pmc_traffic_celeba_off.json"))["celeba"]
for k in a:
    if k in b: print("%-44s bytes/launch xcd %6.0f MB  off %6.0f MB" % (k, a[k]["bytes_per_launch"]/1e6, b[k]["bytes_per_launch"]/1e6))
PY
