#!/bin/bash
# backward tail tiles A/B: shard proxy (1/8, 1/4, 1/2 of cfg2) and cfg2 itself, library vs build/ab_*.so; then the GPU tests
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash scripts/gpu_shard_ab.sh
O=gpurun_out/shard_ab
for r in 1 2; do for lib in $R/tangram_amd/csrc/libtangram_hip.so $R/build/ab_*.so; do
  [ -f $lib ] || continue; n=$(basename $lib .so)
  TANGRAM_AMD_LIB=$lib timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-alt > $O/cfg2_${n}_r$r.json 2> $O/cfg2_${n}_r$r.err
  python - $O/cfg2_${n}_r$r.json $n <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("cfg2", sys.argv[2], round(d["ms_per_step"],3), {x["name"]:round(x["avg_ms"],3) for x in d["kernels"] if x["avg_ms"]>0.1})
PY
done; done
