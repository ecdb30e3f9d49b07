cd /tmp && export TMPDIR=/tmp
for d in ${DBGS:-0 7 15 16}; do
  VY_SKINNY_DBG=$d rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/sk_$d -o s -- python3 $GRAFT_REPO_ROOT/tools/bench_decode_step.py --iters 10 > /dev/null 2>&1
done
