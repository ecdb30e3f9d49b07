import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes, torch
from vyomai_amd import _lib
lib = _lib.load()
torch.zeros(1, device="cuda")
print("occupancy (workgroups per CU) dh=64:", lib.vy_debug_attn_occupancy(64), " dh=128:", lib.vy_debug_attn_occupancy(128))
print(torch.cuda.get_device_properties(0))
