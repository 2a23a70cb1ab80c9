#!/bin/bash
# PMC counter passes (one rocprofv3 run per counter group, --pmc only: no trace domains) for bench.py.
# usage: gpu_pmc.sh tag precision [traffic|all] [keep] ["extra bench args"]   ("traffic": only the groups profiles/pmc_traffic.json is built from)
TAG=${1:-pmc}; P=${2:-bf16x3}; MODE=${3:-all}; KEEP=${4:-}; EXTRA=${5:-}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
[ -z "$KEEP" ] && rm -rf $R/gpurun_out/*
mkdir -p $O
cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1
i=0
while read -r GROUP; do
  i=$((i+1))
  if [ "$MODE" = traffic ]; then case "$GROUP" in SQ_VALU_MFMA*|TCC_HIT*|FETCH_SIZE|WRITE_SIZE|GRBM*) ;; *) continue;; esac; fi
  timeout 300 rocprofv3 --pmc $GROUP --output-format csv -d $O/g$i -o p -- python $R/bench.py --steps 3 --warmup 1 --precision $P --no-cpu-baseline --no-alt $EXTRA > $O/g$i.log 2>&1
  echo "group $i [$GROUP] rc=$?"
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM
SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
FETCH_SIZE
WRITE_SIZE
GRBM_GUI_ACTIVE GRBM_COUNT
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES SQ_CYCLES
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
GROUPS
cd $R
python - $O <<'PY'
import csv,glob,sys,collections,os
O=sys.argv[1]
for f in sorted(glob.glob(O+"/g*/**/*counter_collection.csv",recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k=row.get("Kernel_Name","")
            if not k.startswith(("void tg_","tg_")): continue
            k=k.split("(")[0].replace("void ","")
            agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); 
            cnt[(k,row["Counter_Name"])]+=1
    out=f.replace("counter_collection.csv","summary.txt")
    with open(out,"w") as o:
        for k in agg:
            for c,v in agg[k].items():
                line="%-44s %-28s avg_per_dispatch=%.6g n=%d"%(k,c,v/cnt[(k,c)],cnt[(k,c)])
                o.write(line+"\n")
                if "bwd" in k or "fwd" in k or "rowpass" in k or "adam_update" in k: print(line)
    os.remove(f)
PY
find $O -size +512k -delete; du -sh $R/gpurun_out; tail -3 $O/g3.log
