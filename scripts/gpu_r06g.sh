#!/bin/bash
# round 6, call G: the forward's work decomposition where the equal-ranges choice was a single partly filled round -- ONE exact round of
# stream-K pieces (the new automatic choice) against the old choice forced (+eqN), on shards and on two single-GPU shapes
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06g
rm -rf $R/gpurun_out/*; mkdir -p $O; cd $R; export PYTHONUNBUFFERED=1
timeout 900 python scripts/bench_shard_proxy.py 8_bf16x3_rccl 8_bf16_rccl 4_bf16x3_rccl 2_bf16x3_rccl 8,bf16x3,rccl+eq12 8,bf16,rccl+eq12 8,fp32,rccl 8,fp32,rccl+eq12 4,bf16x3,rccl+eq6 2,bf16x3,rccl+eq3 > $O/shard.json 2> $O/shard.err; echo "rc=$?"; tail -2 $O/shard.err
python - $O <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]+"/shard.json").read().strip().splitlines()[-1])
for k,v in d.items(): print(k, "%.4f ms"%v["ms_per_step"], "loss %.9f"%v["main_loss"], {n:v["kernels_us"][n] for n in ("tg_fwd_kernel","tg_ghat_reduce","tg_bwd_kernel")})
PY
for SH in "26431,249,9852 6" "8000,500,5000 6" "30000,1000,3300 4"; do set -- $SH
  for P in bf16x3 bf16; do
    A=$(timeout 300 python bench.py --shape $1 --precision $P --steps 100 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f it/s fwd %.1f us'%(d['value'], 1e3*[k['avg_ms'] for k in d['kernels'] if k['name']=='tg_fwd_kernel'][0]))")
    B=$(timeout 300 python bench.py --shape $1 --precision $P --splits $2 --steps 100 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f it/s fwd %.1f us'%(d['value'], 1e3*[k['avg_ms'] for k in d['kernels'] if k['name']=='tg_fwd_kernel'][0]))")
    echo "$1 $P: automatic $A | $2 equal ranges per tile $B"
  done
done
