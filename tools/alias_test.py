import torch, ctypes
hip = ctypes.CDLL("libamdhip64.so")
p = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(p), 8 * 16) == 0
class Dev:
    def __init__(self, ptr, n): self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}
t = torch.as_tensor(Dev(p.value, 16), device="cuda")
print(t.data_ptr() == p.value, t.dtype, t.shape)
t.fill_(3.0); t.mul_(2)
import numpy as np
h = np.zeros(16)
assert hip.hipMemcpy(h.ctypes.data_as(ctypes.c_void_p), p, 8 * 16, 2) == 0
print(h[:4])
