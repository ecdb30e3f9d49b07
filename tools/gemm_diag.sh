#!/bin/bash
for v in 0 3 4 5; do
  echo "=== VY_GEMM_VARIANT=$v"
  VY_GEMM_VARIANT=$v python tools/bench_kernels.py 2>&1 | grep -E "out\+res|ffn2"
done
