import torch, time
n = 3840*2160*3
d = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(4)]
h = [torch.empty(n, dtype=torch.uint8, pin_memory=True) for _ in range(4)]
def t(f, reps=20):
    f(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/reps
one = lambda: h[0].copy_(d[0], non_blocking=True)
print("1 stream: %.1f GB/s per copy (%.3f ms)" % (n/t(one)/1e9, t(one)*1e3))
ss = [torch.cuda.Stream() for _ in range(4)]
def multi(k):
    def f():
        for i in range(k):
            with torch.cuda.stream(ss[i]): h[i].copy_(d[i], non_blocking=True)
    return f
for k in (2,3,4):
    dt = t(multi(k)); print("%d streams: %.1f GB/s total (%.3f ms per batch)" % (k, k*n/dt/1e9, dt*1e3))
# split one buffer in halves on 2 streams
def halves():
    with torch.cuda.stream(ss[0]): h[0][:n//2].copy_(d[0][:n//2], non_blocking=True)
    with torch.cuda.stream(ss[1]): h[0][n//2:].copy_(d[0][n//2:], non_blocking=True)
dt = t(halves); print("one buffer as two halves on 2 streams: %.1f GB/s (%.3f ms)" % (n/dt/1e9, dt*1e3))
# kernel copy into mapped pinned memory via a device-side view of the host tensor
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
ptr = ctypes.c_void_p()
rc = hip.hipHostGetDevicePointer(ctypes.byref(ptr), ctypes.c_void_p(h[1].data_ptr()), 0)
print("hipHostGetDevicePointer rc", rc, hex(ptr.value or 0), hex(h[1].data_ptr()))
