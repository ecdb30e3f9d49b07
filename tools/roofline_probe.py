"""The four forward GEMM launches of one layer (the kernel bench.py's `roofline` prices), alone,
for rocprofv3 --pmc / --kernel-trace runs.  Same shapes and fusion as bench.roofline_probe."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vyomai_amd import EncoderConfig
cfg = EncoderConfig(num_hidden_layers=12, max_position_embeddings=1024, hidden_dropout_prob=0.0)
r = bench.roofline_probe(cfg, 32, 512, torch.device("cuda", 0))
print(r)
