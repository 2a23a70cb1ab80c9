#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06c
rm -rf $R/gpurun_out/*; mkdir -p $O; cd $R; export PYTHONUNBUFFERED=1
timeout 600 python scripts/probes/peer_fused_diff.py > $O/diff.log 2>&1; echo "rc=$?"; grep -v "Gloo\|amdgpu.ids\|^\[W" $O/diff.log | tail -20
