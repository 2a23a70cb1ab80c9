#!/bin/bash
# Same-box A/B of kernel variants: the in-tree library vs every build/ab_*.so (scripts/build_variant.sh), interleaved rounds.
# usage: gpu_ab.sh tag "precisions" rounds ["extra bench args"] [keep]     (keep: do not wipe gpurun_out first)
TAG=${1:-ab}; PRECS=${2:-"bf16x3"}; ROUNDS=${3:-2}; EXTRA=${4:-}; KEEP=${5:-}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG
[ -z "$KEEP" ] && rm -rf $R/gpurun_out/*
mkdir -p $O; cd $R; export PYTHONUNBUFFERED=1
for r in $(seq 1 $ROUNDS); do
 for lib in $R/tangram_amd/csrc/libtangram_hip.so $R/build/ab_*.so; do
  [ -f $lib ] || continue; n=$(basename $lib .so)
  for P in $PRECS; do
    timeout 600 python scripts/with_lib.py $lib bench.py --steps 40 --warmup 5 --precision $P --no-cpu-baseline --no-alt $EXTRA > $O/${n}_${P}_r$r.json 2> $O/${n}_${P}_r$r.err || echo "FAIL $n $P"
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
with open(sys.argv[1]+"/summary.txt","w") as o:
    for key,v in rows.items():
        line="%-40s"%key+" | ".join("step %.3f fwd %.3f bwd %.3f adam %.3f"%x[:4] for x in v)+"  loss %.6f"%v[0][4]
        print(line); o.write(line+"\n")
PY
