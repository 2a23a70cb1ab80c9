#!/bin/bash
# round 6, call F: how tight can the small-case bound on max|dP| of plain bf16 be?  The bf16 cases of the GPU suite under shrinking bounds.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06f
rm -rf $R/gpurun_out/*; mkdir -p $O; cd $R; export PYTHONUNBUFFERED=1
for B in 2e-2 1e-2 5e-3 2e-3 1e-3; do
  TG_TOL_BF16_P=$B timeout 900 python -m pytest tests -m gpu -q --timeout 600 -k "bf16 and not bf16x3 and not full_size and not cfg4" > $O/bf16_$B.log 2>&1
  echo "bound $B: rc=$? $(tail -1 $O/bf16_$B.log)"; grep "^FAILED" $O/bf16_$B.log | cut -c1-200 | head -8
done
