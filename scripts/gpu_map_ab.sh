#!/bin/bash
# dense vs padded XCD tile map of the backward GEMM: library vs build/ab_*.so on several shapes + the shard proxy
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/map_ab; rm -rf gpurun_out/*; mkdir -p $O
for lib in $R/tangram_amd/csrc/libtangram_hip.so $R/build/ab_*.so; do
 n=$(basename $lib .so)
 for SH in 30000,1000,10000 10000,1000,10000 20000,2000,3000 30000,1000,1000; do
  for PIN in 256 128; do
   TANGRAM_AMD_BWD_TILE=$PIN TANGRAM_AMD_LIB=$lib timeout 200 python bench.py --shape $SH --steps 30 --warmup 5 --no-cpu-baseline --no-alt > $O/${n}_${SH}_$PIN.json 2> $O/${n}_${SH}_$PIN.err || echo "FAIL $SH $PIN"
   python - $O/${n}_${SH}_$PIN.json $n $SH $PIN <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2],sys.argv[3],"bwd",sys.argv[4],"ms/step %.4f"%d["ms_per_step"],{x["name"]:round(x["avg_ms"],4) for x in d["kernels"] if "bwd" in x["name"]}, "loss %.6f"%d["last_main_loss"])
except Exception as e: print("parse fail",sys.argv[1:],e)
PY
  done
 done
 for PIN in 256 128; do
  TANGRAM_AMD_BWD_TILE=$PIN TANGRAM_AMD_LIB=$lib timeout 300 python scripts/bench_shard_proxy.py > $O/${n}_shard_$PIN.json 2> $O/${n}_shard_$PIN.err
  python - $O/${n}_shard_$PIN.json $n $PIN <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
for k,v in d.items(): print(sys.argv[2],"shard bwd",sys.argv[3],k,round(v["ms_per_step"],3),v["kernels_us"].get("tg_bwd_kernel"))
PY
 done
done
