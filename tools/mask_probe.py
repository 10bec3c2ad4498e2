"""Which CU masks does hipExtStreamCreateWithCUMask accept here?  (streams are destroyed again: hardware queues are few)"""
import ctypes as C
import torch
torch.zeros(1, device="cuda")
def lib():
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return line.split()[-1]
hip = C.CDLL(lib())
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipStreamDestroy.argtypes = [C.c_void_p]
hip.hipGetErrorString.restype = C.c_char_p
pats = {"all": [0xffffffff] * 8, "low16 per word": [0xffff] * 8, "0x55555555": [0x55555555] * 8, "0x01010101": [0x01010101] * 8,
        "0xfefefefe": [0xfefefefe] * 8, "first word only": [0xffffffff] + [0] * 7, "last 7 words": [0] + [0xffffffff] * 7,
        "0x0000000f": [0xf] * 8, "0xfffffff0": [0xfffffff0] * 8}
for name, words in pats.items():
    w = (C.c_uint32 * 8)(*words)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, w)
    print(f"{name:18s} rc={rc} {hip.hipGetErrorString(rc).decode()}", flush=True)
    hip.hipGetLastError()
    if rc == 0:
        hip.hipStreamDestroy(s)
