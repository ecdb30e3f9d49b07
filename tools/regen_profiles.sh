#!/bin/bash
# Regenerate the rocprofv3 evidence under profiles/ for round $1 (e.g. r02).  Run on the GPU box:
#   gpurun -- 'bash tools/regen_profiles.sh r02'
# then, back in the build container:  python tools/collect_profiles.py r02
# Counters are collected in their own passes (kernel-trace / stats and --pmc are never combined; FETCH_SIZE and
# WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md, rocprofv3 PMC slots).
set -e
R=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/roofline -o rp -- python3 $ROOT/tools/roofline_probe.py > $OUT/roofline.log 2>&1
echo "roofline probe traced"
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- python3 $ROOT/tools/roofline_probe.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- python3 $ROOT/tools/roofline_probe.py > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_hit -o p -- python3 $ROOT/tools/roofline_probe.py > $OUT/pmc_hit.log 2>&1
echo "pmc passes done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/decode -o dec -- python3 $ROOT/tools/bench_decode_step.py --iters 20 > $OUT/decode.log 2>&1
echo "decode step traced"
python3 $ROOT/tools/decode_timeline.py --layers 12 > $OUT/decode_timeline.log 2>&1
echo "decode timeline written"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/paligemma -o pg -- python3 $ROOT/tools/bench_paligemma.py > $OUT/paligemma.log 2>&1
echo "configs[4] decode traced"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prefill -o pf -- python3 $ROOT/tools/bench_paligemma_prefill.py 6 > $OUT/prefill.log 2>&1
echo "configs[4] prefill traced"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/vlm -o v -- python3 $ROOT/tools/bench_vlm_training.py > $OUT/vlm.log 2>&1
echo "configs[3] training traced"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -o t -- python3 $ROOT/tools/train_only.py 10 > $OUT/train.log 2>&1
echo "configs[1] training steps traced"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o b -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/bench.log 2>&1
echo "bench traced"
# the raw per-launch traces are large and not needed (the stats tables are what profiles/ keeps)
find $OUT -name "*kernel_trace.csv" -delete
