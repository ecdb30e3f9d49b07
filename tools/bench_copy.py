"""The HBM ceiling probe behind bench.py's measured_ceilings (vy_debug_copy), per mode / grid (each combination in its own
process: the knobs are read once):   VY_COPY_MODE=3 VY_COPY_WGS=2048 python tools/bench_copy.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vyomai_amd import _lib
lib = _lib.load()
lib.vy_debug_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
n = 1 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a); a.zero_()
st = torch.cuda.current_stream()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    lib.vy_debug_copy(a.data_ptr(), b.data_ptr(), n, st.cuda_stream)
torch.cuda.synchronize(); s.record()
for _ in range(10):
    lib.vy_debug_copy(a.data_ptr(), b.data_ptr(), n, st.cuda_stream)
e.record(); torch.cuda.synchronize()
t = s.elapsed_time(e) / 10
print(f"mode {os.environ.get('VY_COPY_MODE', '3')} wgs {os.environ.get('VY_COPY_WGS', '2048')}: {2 * n / t * 1e-9:.3f} TB/s (read + write)")
