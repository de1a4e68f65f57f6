"""Known-byte workload for calibrating rocprofv3 FETCH_SIZE / WRITE_SIZE with this library's access shape.
Run under `rocprofv3 --pmc FETCH_SIZE ...` and `--pmc WRITE_SIZE ...` (separate passes); each k_calib_copy launch
reads and writes exactly 8*n bytes (n = 2**27 doubles = 1 GiB each way, far beyond the 256 MiB Infinity Cache)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from miniweatherml_amd import capi

n = 2 ** 27
a = torch.rand(n, dtype=torch.float64, device="cuda")
b = torch.empty_like(a)
L = capi.lib()
for _ in range(5):
    capi.check(L.mw_calib_copy(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), n, None))
torch.cuda.synchronize()
assert torch.equal(a, b)
print("bytes_each_way", 8 * n)
