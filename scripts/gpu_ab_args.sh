#!/bin/bash
# same-box A/B over several bench argument sets: gpu_ab_args.sh tag "args1" "args2" ...   (libs: current + build/ab_*.so)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG
rm -rf $R/gpurun_out/*; mkdir -p $O; cd $R
i=0
for A in "$@"; do i=$((i+1)); for r in 1 2; do for lib in $R/tangram_amd/csrc/libtangram_hip.so $R/build/ab_*.so; do
  [ -f $lib ] || continue; n=$(basename $lib .so)
  TANGRAM_AMD_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --no-alt $A > $O/${n}_a${i}_r$r.json 2> $O/${n}_a${i}_r$r.err || echo FAIL
  python - $O/${n}_a${i}_r$r.json "$n | $A" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-70s step %.4f ms  "%(sys.argv[2],d["ms_per_step"]), {x["name"][3:]:round(x["avg_ms"]*1000,1) for x in d["kernels"]})
PY
done; done; done
