import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
order = sys.argv[1] if len(sys.argv) > 1 else "lib_first"
def maps():
    return sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'hip64' in l or 'hsa-runtime' in l or 'rccl' in l))
if order == "torch_first":
    import torch; torch.zeros(1, device="cuda"); print("torch ok")
from miniweatherml_amd import capi
L = capi.lib()
print("mw_device_count:", L.mw_device_count())
print(maps())
if order != "torch_first":
    import torch; x = torch.zeros(1, device="cuda"); print("torch ok", torch.cuda.device_count())
    print("mw_device_count after torch:", L.mw_device_count())
    print(maps())
hip = ctypes.CDLL("libamdhip64.so.7")
n = ctypes.c_int(-1); rc = hip.hipGetDeviceCount(ctypes.byref(n)); print("direct hipGetDeviceCount rc", rc, n.value)
