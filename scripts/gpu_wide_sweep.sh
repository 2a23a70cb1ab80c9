#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/wide_sweep; rm -rf gpurun_out/*; mkdir -p $O
for SH in 30000,1000,1000 30000,1000,500 10000,1000,10000 20000,2000,3000 5000,1000,2000 8000,500,4000; do
 for lib in $R/tangram_amd/csrc/libtangram_hip.so $R/build/ab_nowide.so; do n=$(basename $lib .so)
   TANGRAM_AMD_LIB=$lib timeout 200 python bench.py --shape $SH --steps 30 --warmup 5 --no-cpu-baseline --no-alt > $O/${n}_$SH.json 2> $O/${n}_$SH.err || echo FAIL
   python - $O/${n}_$SH.json $n $SH <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k={x["name"]:round(x["avg_ms"],4) for x in d["kernels"]}
print(sys.argv[3],sys.argv[2],"ms/step %.4f"%d["ms_per_step"],"fwd",k.get("tg_fwd_kernel"),"reduce",k.get("tg_ghat_reduce"))
PY
 done
done
for lib in $R/tangram_amd/csrc/libtangram_hip.so $R/build/ab_nowide.so; do n=$(basename $lib .so)
  TANGRAM_AMD_LIB=$lib timeout 300 python scripts/bench_shard_proxy.py > $O/${n}_shard.json 2> $O/${n}_shard.err
  python - $O/${n}_shard.json $n <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
for k,v in d.items(): print(sys.argv[2],"shard",k,round(v["ms_per_step"],3),"fwd",v["kernels_us"].get("tg_fwd_kernel"))
PY
done
