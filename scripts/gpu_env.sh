#!/bin/bash
# same-box comparison of one env knob: gpu_env.sh tag VAR "values" "precisions" [rounds]
TAG=$1; VAR=$2; VALS=$3; PRECS=${4:-"bf16x3"}; ROUNDS=${5:-2}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG
rm -rf $R/gpurun_out/*; mkdir -p $O; cd $R
for r in $(seq 1 $ROUNDS); do for v in $VALS; do for P in $PRECS; do
  env $VAR=$v timeout 600 python bench.py --steps 20 --warmup 4 --precision $P --no-cpu-baseline --no-alt > $O/${VAR}_${v}_${P}_r$r.json 2> $O/${VAR}_${v}_${P}_r$r.err || echo FAIL $v $P
done; done; done
python - $O <<'PY'
import json,glob,sys,os,collections
rows=collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception: print("parse fail",f); continue
    key=os.path.basename(f)[:-5].rsplit("_r",1)[0]
    k={x["name"]:x["avg_ms"] for x in d["kernels"]}
    rows[key].append((d["ms_per_step"],k.get("tg_fwd_kernel",0),k.get("tg_bwd_kernel",0),k.get("tg_adam_update",0)+k.get("tg_adam_rowpass",0)))
for key,v in rows.items():
    print("%-40s"%key," | ".join("step %.3f fwd %.3f bwd %.3f adam %.3f"%x for x in v))
PY
