#!/bin/bash
# automatic geometry (layout rule + tg_tune_bwd) vs pinned backward tiles; shard proxy with pinned backward tiles
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/tile_sweep3; rm -rf gpurun_out/*; mkdir -p $O
for SH in 30000,1000,10000 10000,1000,10000 20000,2000,3000 30000,1000,1000 30000,1000,500 20000,1000,324 5000,1000,2000; do
 for PIN in auto 128 256; do
   if [ $PIN = auto ]; then unset TANGRAM_AMD_BWD_TILE; else export TANGRAM_AMD_BWD_TILE=$PIN; fi
   timeout 200 python bench.py --shape $SH --steps 30 --warmup 5 --no-cpu-baseline --no-alt > $O/${SH}_$PIN.json 2> $O/${SH}_$PIN.err || echo "FAIL $SH $PIN"
   python - $O/${SH}_$PIN.json $SH $PIN <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2],"bwd",sys.argv[3],"ms/step %.4f"%d["ms_per_step"],{x["name"]:round(x["avg_ms"],4) for x in d["kernels"] if x["avg_ms"]>0.03})
except Exception as e: print("parse fail",sys.argv[1:],e)
PY
 done
done
for PIN in 256 128; do
  TANGRAM_AMD_BWD_TILE=$PIN timeout 300 python scripts/bench_shard_proxy.py > $O/shard_$PIN.json 2> $O/shard_$PIN.err
  python - $O/shard_$PIN.json $PIN <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
for k,v in d.items(): print("shard bwd",sys.argv[2],k,round(v["ms_per_step"],3),v["kernels_us"].get("tg_bwd_kernel"),v["kernels_us"].get("tg_rowsum_parts"))
PY
done
