#!/bin/bash
for cfg in "0 0 0" "0 1 0" "0 0 512" "0 0 768" "0 0 256" "0 1 768"; do
  set -- $cfg
  echo "== variant=$1 diag=$2 target=$3"
  VY_WGRAD_VARIANT=$1 VY_WGRAD_DIAG=$2 VY_WGRAD_TARGET=$3 python tools/bench_wgrad.py 2>&1 | grep wgrad
done
