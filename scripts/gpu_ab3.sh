#!/bin/bash
# one call: (1) bench A/B of the library vs build/ab_*.so (names matching $2), (2) shard-proxy A/B (names matching $3), (3) pytest -m gpu
TAG=${1:-ab}; BENCH_RE=${2:-.}; SHARD_RE=${3:-NONE}; PRECS=${4:-"bf16x3 bf16"}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG
rm -rf $R/gpurun_out/*; mkdir -p $O; cd $R; export PYTHONUNBUFFERED=1
for r in 1 2; do
 for lib in $R/tangram_amd/csrc/libtangram_hip.so $R/build/ab_*.so; do
  [ -f $lib ] || continue; n=$(basename $lib .so)
  if [ "$n" != libtangram_hip ] && ! echo $n | grep -qE "$BENCH_RE"; then continue; fi
  for P in $PRECS; do
    TANGRAM_AMD_LIB=$lib timeout 600 python bench.py --steps 40 --warmup 5 --precision $P --no-cpu-baseline --no-alt > $O/${n}_${P}_r$r.json 2> $O/${n}_${P}_r$r.err || echo "FAIL $n $P"
  done
 done
done
python - $O <<'PY'
import json,glob,sys,os,collections
rows=collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1]+"/*_r?.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print("parse fail",f); continue
    b=os.path.basename(f)[:-5]; key=b.rsplit("_r",1)[0]
    k={x["name"]:x["avg_ms"] for x in d["kernels"]}
    rows[key].append((d["ms_per_step"],k.get("tg_fwd_kernel",0),k.get("tg_bwd_kernel",0),(k.get("tg_adam_update",0)+k.get("tg_adam_rowpass",0)),d["last_main_loss"]))
for key,v in rows.items():
    print("%-36s"%key," | ".join("step %.3f fwd %.3f bwd %.3f adam %.3f"%x[:4] for x in v), " loss %.6f"%v[0][4])
PY
if [ "$SHARD_RE" != NONE ]; then
 for r in 1 2; do for lib in $R/tangram_amd/csrc/libtangram_hip.so $R/build/ab_*.so; do
  [ -f $lib ] || continue; n=$(basename $lib .so)
  if [ "$n" != libtangram_hip ] && ! echo $n | grep -qE "$SHARD_RE"; then continue; fi
  TANGRAM_AMD_LIB=$lib timeout 600 python scripts/bench_shard_proxy.py > $O/shard_${n}_r$r.jsonl 2> $O/shard_${n}_r$r.err
  python - $O/shard_${n}_r$r.jsonl $n <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    for k,v in d.items(): print(sys.argv[2], k, round(v["ms_per_step"],3), "enq", round(v["host_enqueue_ms_per_step"],3), v["kernels_us"])
except Exception as e: print(sys.argv[2], "FAIL", e)
PY
 done; done
fi
S=$SECONDS; timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$? in $((SECONDS-S)) s"; tail -3 $O/pytest_gpu.log
du -sh $R/gpurun_out | tail -1
