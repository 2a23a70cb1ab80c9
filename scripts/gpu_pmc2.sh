#!/bin/bash
# extra PMC groups (stall-side counters) for the GEMM kernels: gpu_pmc2.sh tag precision
TAG=${1:-pmcx}; P=${2:-bf16x3}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG
rm -rf $R/gpurun_out/*; mkdir -p $O
cd /tmp && export TMPDIR=/tmp PYTHONUNBUFFERED=1
i=0
while read -r GROUP; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $GROUP --output-format csv -d $O/g$i -o p -- python $R/bench.py --steps 3 --warmup 1 --precision $P --no-cpu-baseline --no-alt > $O/g$i.log 2>&1
  echo "group $i [$GROUP] rc=$?"
done <<'GROUPS'
SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL
SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL
SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
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
            if not ("tg_fwd_kernel" in k or "tg_bwd_kernel" in k or "tg_adam_rowpass" in k): continue
            k=k.split("<")[0].replace("void ","")
            agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[(k,row["Counter_Name"])]+=1
    out=f.replace("counter_collection.csv","summary.txt")
    with open(out,"w") as o:
        for k in agg:
            for c,v in agg[k].items():
                line="%-18s %-30s avg_per_dispatch=%.6g n=%d"%(k,c,v/cnt[(k,c)],cnt[(k,c)])
                o.write(line+"\n"); print(line)
    os.remove(f)
PY
find $O -size +512k -delete
