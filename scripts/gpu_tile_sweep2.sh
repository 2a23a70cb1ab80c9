#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/tile_sweep2; rm -rf gpurun_out/*; mkdir -p $O
for SH in 30000,1000,10000 30000,1000,2500 30000,2000,10000 10000,1000,10000; do
 for P in bf16x3 bf16; do
  for T in 128 256; do
   timeout 200 python bench.py --shape $SH --tile $T --precision $P --steps 30 --warmup 5 --no-cpu-baseline --no-alt > $O/${SH}_${P}_$T.json 2> $O/${SH}_${P}_$T.err || echo "FAIL $SH $T"
   python - $O/${SH}_${P}_$T.json $SH $P $T <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2],sys.argv[3],"tile",sys.argv[4],"ms/step %.4f"%d["ms_per_step"],{x["name"]:round(x["avg_ms"],4) for x in d["kernels"] if x["avg_ms"]>0.03})
except Exception as e: print("parse fail",sys.argv[1:],e)
PY
  done
 done
done
