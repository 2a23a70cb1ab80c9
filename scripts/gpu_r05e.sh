#!/bin/bash
# round 5, call E: clusters-mode kernels after the diet (rocprofv3 stats of one fold / 16 folds, stepping rates), peer tests with the
# strict fine-grained mailbox, and the fuzz sweeps (160 random production-geometry problems, 50 sharded) on the new update kernels.
TAG=${1:-r05e}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG
rm -rf $R/gpurun_out/*; mkdir -p $O; cd $R; export PYTHONUNBUFFERED=1
echo "== peer"; timeout 600 python -m pytest tests/test_gpu_baseline_shapes.py -m gpu -q --timeout 600 -k "peer_transport" > $O/pytest_peer.log 2>&1; echo "rc=$?"; tail -3 $O/pytest_peer.log
echo "== batched rates"; timeout 600 python scripts/bench_batched.py --no-e2e --batches 8,16 > $O/batched.json 2> $O/batched.err; echo "rc=$?"; tail -c 600 $O/batched.json
cd /tmp && export TMPDIR=/tmp
for B in 1 16; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b$B -o r -- python $R/scripts/prof_batch.py $B > $O/prof_b$B.log 2>&1; echo "rocprof B=$B rc=$?"
  head -7 $O/prof_b$B/*kernel_stats.csv | cut -c1-130
done
cd $R; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -size +1M -delete
echo "== fuzz"; TG_FUZZ_SEEDS=160 timeout 1200 python -m pytest tests/test_gpu_production_tiles.py -m gpu -q --timeout 900 -k "random_production" > $O/pytest_fuzz.log 2>&1; echo "rc=$?"; tail -3 $O/pytest_fuzz.log
du -sh $R/gpurun_out
