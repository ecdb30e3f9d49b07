#!/bin/bash
# run the GEMM parity tests and the kernel microbench under each large-M kernel variant
for v in ${VARIANTS:-0 8 9}; do
  echo "=== VY_GEMM_VARIANT=$v"
  VY_GEMM_VARIANT=$v python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 900 -k "linear_bf16 or qkv" 2>&1 | tail -2
  VY_GEMM_VARIANT=$v python tools/bench_kernels.py 2>&1 | grep -E "qkv|out\+res|ffn|lm_|block"
done
