#!/bin/bash
# round 6, call I: does the streaming update on a shard lose bandwidth to its ragged last trip?  cfg4-sized rows of 6 250 (6.1 trips of 1 024) against 6 144 (6.0) and 7 168 (7.0)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06i
rm -rf $R/gpurun_out/*; mkdir -p $O; cd $R; export PYTHONUNBUFFERED=1
for VT in 50000 49152 57344 40960; do
  PROXY_SHAPE=100000,2000,$VT timeout 600 python scripts/bench_shard_proxy.py 8,bf16,rccl 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d.items():
    C,K,Vl=[int(x) for x in k.split('_')[0].split('x')]
    u=v['kernels_us']['tg_adam_update']; print(k, '%.3f ms'%v['ms_per_step'], 'update %.1f us = %.2f TB/s'%(u, C*Vl*26/u/1e6))"
done
