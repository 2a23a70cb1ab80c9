#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_batch
rm -rf $R/gpurun_out/*; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for B in 1 8; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b$B -o r -- python $R/scripts/prof_batch.py $B > $O/b$B.log 2>&1; echo "B=$B rc=$?"
find $O/b$B -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
head -9 $O/b$B/*kernel_stats.csv | cut -c1-140
done
