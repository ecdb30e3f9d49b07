cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R && python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "linear_bf16 or qkv" 2>&1 | tail -2
python tools/exp_gemm.py -1 2>&1 | grep "variant  -1:"
VY_GEMM_WT_STORE=0 python tools/exp_gemm.py -1 2>&1 | grep "variant  -1:"
cd /tmp
for mode in wt; do
  rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/band_$mode/f -o p -- python3 $R/tools/roofline_probe.py > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/band_$mode/w -o p -- python3 $R/tools/roofline_probe.py > /dev/null 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/band_$mode/h -o p -- python3 $R/tools/roofline_probe.py > /dev/null 2>&1
done
