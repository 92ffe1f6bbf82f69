"""What do hipMalloc / hipFree cost on this box?  (Setup routines allocate dozens of temporaries.)"""
import ctypes
import time

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipFree.argtypes = [ctypes.c_void_p]
hip.hipDeviceSynchronize()
for size in (4096, 1 << 20, 4 << 20, 16 << 20, 64 << 20, 256 << 20):
    ptrs = []
    p = ctypes.c_void_p()
    hip.hipMalloc(ctypes.byref(p), size)
    hip.hipFree(p)
    t0 = time.perf_counter()
    for _ in range(20):
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), size) == 0
        ptrs.append(p)
    t1 = time.perf_counter()
    for p in ptrs:
        hip.hipFree(p)
    t2 = time.perf_counter()
    print(f"{size >> 10:8d} KiB: hipMalloc {(t1 - t0) / 20 * 1e6:8.1f} us  hipFree {(t2 - t1) / 20 * 1e6:8.1f} us")
