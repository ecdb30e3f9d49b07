#!/bin/bash
for v in 0 1; do echo "== VY_WGRAD_VARIANT=$v"; VY_WGRAD_VARIANT=$v python tools/bench_wgrad.py 2>&1 | grep -v amdgpu; done
python -m pytest tests/test_bwd_kernels_gpu.py tests/test_training_gpu.py -m gpu -q -x 2>&1 | tail -2
VY_WGRAD_VARIANT=1 python -m pytest tests/test_bwd_kernels_gpu.py -m gpu -q -x -k wgrad 2>&1 | tail -2
