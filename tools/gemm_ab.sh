#!/bin/bash
for v in ${VARIANTS:-0 5 9}; do
  echo "=== VY_GEMM_VARIANT=$v"
  VY_GEMM_VARIANT=$v python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "linear_bf16" 2>&1 | tail -1
  VY_GEMM_VARIANT=$v python tools/bench_kernels.py 2>&1 | grep -E "qkv\(|out\+res|ffn"
done
