#!/bin/bash
# round 5, call A: the slimmed update kernels.  Same-box A/B against round 4's library (build/ab_r04.so), the CU-mask probe of the
# review's gate (scripts/probes/cumask_step.py), SQ_INSTS_VALU of the update kernel, then the whole GPU suite.
# usage: gpu_r05a.sh tag [stages]      stages: any of  ab probe pmc tests  (default: all)
TAG=${1:-r05a}; STAGES=${2:-"ab probe pmc tests"}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG
rm -rf $R/gpurun_out/*; mkdir -p $O; cd $R; export PYTHONUNBUFFERED=1
(rocminfo | grep -E "Marketing|gfx" | head -4; nproc; free -g | head -2) > $O/env.log 2>&1
for S in $STAGES; do
case $S in
ab)
  for r in 1 2; do
   for lib in $R/tangram_amd/csrc/libtangram_hip.so $R/build/ab_*.so; do
    [ -f $lib ] || continue; n=$(basename $lib .so)
    timeout 300 python scripts/with_lib.py $lib bench.py --steps 40 --warmup 5 --precision bf16x3 --no-cpu-baseline --no-alt > $O/${n}_r$r.json 2> $O/${n}_r$r.err || echo "FAIL $n"
   done
  done
  python - $O <<'PY'
import json,glob,sys,os
for f in sorted(glob.glob(sys.argv[1]+"/*_r?.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print("parse fail",f); continue
    k={x["name"]:round(x["avg_ms"],4) for x in d["kernels"] if x["avg_ms"]>0.004}
    print(os.path.basename(f), "%.1f it/s %.3f ms"%(d["value"],d["ms_per_step"]), k, "loss %.7f"%d["last_main_loss"])
PY
  ;;
probe)
  timeout 600 python scripts/probes/cumask_step.py > $O/cumask_step.txt 2> $O/cumask_step.err; echo "probe rc=$?"; cat $O/cumask_step.txt | head -12
  ;;
pmc)
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc -o p -- python $R/bench.py --steps 3 --warmup 1 --precision bf16x3 --no-cpu-baseline --no-alt > $O/pmc.log 2>&1
  echo "pmc rc=$?"; cd $R
  python - $O <<'PY'
import csv,glob,sys,collections
for f in glob.glob(sys.argv[1]+"/pmc/**/*counter_collection.csv",recursive=True):
    agg=collections.defaultdict(float); cnt=collections.Counter()
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"]
        if "tg_" not in k: continue
        k=k.split("(")[0].replace("void ","")
        agg[(k,row["Counter_Name"])]+=float(row["Counter_Value"]); cnt[(k,row["Counter_Name"])]+=1
    with open(sys.argv[1]+"/pmc_summary.txt","w") as o:
        for (k,c),v in sorted(agg.items()):
            line="%-50s %-26s avg_per_dispatch=%.6g n=%d"%(k,c,v/cnt[(k,c)],cnt[(k,c)]); o.write(line+"\n")
            if c=="SQ_INSTS_VALU" and ("rowpass" in k or "adam" in k): print(line)
PY
  find $O/pmc -name "*.csv" -size +256k -delete
  ;;
tests)
  timeout 2400 python -m pytest tests -m gpu -q --timeout 1200 -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu.log
  ;;
esac
done
du -sh $R/gpurun_out
