cd /tmp && export TMPDIR=/tmp
for c in 640 641 644 650 672 700; do
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/cap_$c -o s -- python3 $GRAFT_REPO_ROOT/tools/bench_decode_step.py --iters 10 --cap $c > /dev/null 2>&1
done
