#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/major; rm -rf gpurun_out/*; mkdir -p $O
for r in 1 2; do for MJ in cells spots; do for P in bf16x3 bf16; do
  TANGRAM_AMD_BWD_TILE=256 TANGRAM_AMD_BWD_MAJOR=$MJ timeout 200 python bench.py --precision $P --steps 40 --warmup 5 --no-cpu-baseline --no-alt > $O/${MJ}_${P}_$r.json 2> $O/${MJ}_${P}_$r.err || echo FAIL
  python - $O/${MJ}_${P}_$r.json $MJ $P <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k={x["name"]:round(x["avg_ms"],4) for x in d["kernels"]}
print(sys.argv[2],sys.argv[3],"ms/step %.4f"%d["ms_per_step"],"bwd",k.get("tg_bwd_kernel"),"loss %.6f"%d["last_main_loss"])
PY
done; done; done
