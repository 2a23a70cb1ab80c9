#!/bin/bash
# same-box A/B of the sharded driver (1-rank RCCL group on one GPU): current library vs build/ab_*.so, plus the GPU tests
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rm -rf gpurun_out/*; mkdir -p gpurun_out/shard_ab
for r in 1 2; do for lib in $R/tangram_amd/csrc/libtangram_hip.so $R/build/ab_*.so; do
  [ -f $lib ] || continue; n=$(basename $lib .so)
  TANGRAM_AMD_LIB=$lib timeout 600 python scripts/bench_shard_proxy.py > gpurun_out/shard_ab/${n}_r$r.json 2> gpurun_out/shard_ab/${n}_r$r.err
  python - gpurun_out/shard_ab/${n}_r$r.json $n <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], {k.split("x")[2]: (round(v["ms_per_step"],3), v["kernels_us"].get("tg_bwd_kernel")) for k,v in d.items()})
PY
done; done
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
