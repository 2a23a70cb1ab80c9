#!/bin/bash
# round 5, call G (review item 6): X through the Infinity Cache?  build/ab_xtemporal.so stores and re-reads the backward product with
# ordinary (allocating) accesses instead of non-temporal ones; same-box A/B of the step and FETCH_SIZE / WRITE_SIZE of the kernels.
TAG=${1:-r05g}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG
rm -rf $R/gpurun_out/*; mkdir -p $O; cd $R; export PYTHONUNBUFFERED=1
for r in 1 2; do
 for lib in $R/tangram_amd/csrc/libtangram_hip.so $R/build/ab_xtemporal.so; do
  n=$(basename $lib .so)
  timeout 300 python scripts/with_lib.py $lib bench.py --steps 40 --warmup 5 --precision bf16x3 --no-cpu-baseline --no-alt > $O/${n}_r$r.json 2> $O/${n}_r$r.err || echo "FAIL $n"
 done
done
python - $O <<'PY'
import json,glob,sys,os
for f in sorted(glob.glob(sys.argv[1]+"/*_r?.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print("parse fail",f); continue
    print(os.path.basename(f), "%.1f it/s %.3f ms"%(d["value"],d["ms_per_step"]), {x["name"]:round(x["avg_ms"],4) for x in d["kernels"] if x["avg_ms"]>0.1})
PY
cd /tmp && export TMPDIR=/tmp
for lib in $R/tangram_amd/csrc/libtangram_hip.so $R/build/ab_xtemporal.so; do
  n=$(basename $lib .so)
  for G in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $G --output-format csv -d $O/pmc_${n}_$G -o p -- python $R/scripts/with_lib.py $lib $R/bench.py --steps 3 --warmup 1 --precision bf16x3 --no-cpu-baseline --no-alt > $O/pmc_${n}_$G.log 2>&1
  done
done
cd $R
python - $O <<'PY'
import csv,glob,sys,collections
for f in sorted(glob.glob(sys.argv[1]+"/pmc_*/**/*counter_collection.csv",recursive=True)):
    agg=collections.defaultdict(float); cnt=collections.Counter()
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"]
        if not any(s in k for s in ("rowpass","bwd_kernel","fwd_kernel")): continue
        k=k.split("<")[0].replace("void ","")
        agg[(k,row["Counter_Name"])]+=float(row["Counter_Value"]); cnt[(k,row["Counter_Name"])]+=1
    for (k,c),v in sorted(agg.items()): print(f.split("/")[-3], k, c, "%.4g KiB per launch"%(v/cnt[(k,c)]))
PY
find $O -name "*.csv" -size +256k -delete; du -sh $R/gpurun_out
