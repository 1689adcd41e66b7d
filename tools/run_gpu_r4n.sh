#!/bin/bash
# Round 4, call N: images/s of the weak bench against the batch size around the headline's 256 (is --strong's 2500-row launch
# really more efficient per row than 2560 rows, or is it the step structure?).
TAG=${1:-r4n}; O=gpurun_out/$TAG; mkdir -p $O
for b in 250 256 240 248 264 272 288 320 250 256; do
  python bench.py --no-cpu-baseline --steps 6 --warmup 2 --batch $b --retune > $O/b${b}_$(date +%s).json 2>> $O/err.txt
done
python - $O/*.json <<'PY' | tee $O/summary.txt
import json, sys
for f in sorted(sys.argv[1:], key=lambda p: p.split("_")[-1]):
    d = json.load(open(f))
    ks = " ".join("%s %.1f" % (k["name"], k["avg_us"]) for k in d["kernels"])
    B = d["config"]["batch_per_gpu"]
    print("B %4d rows %5d  %8.2f img/s path %.4f  us/row/iter %.4f | %s" % (B, B * 10, d["value"], d["roofline"]["path_frac"], d["ms_per_step"] * 1e3 / 199.5 / (B * 10), ks))
PY
