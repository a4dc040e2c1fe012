"""How long does creating a HIP stream take (with / without a priority), first and later times in a process?"""
import ctypes, time
hip = ctypes.CDLL("libamdhip64.so")
hip.hipSetDevice(0)
p = ctypes.c_void_p()
hip.hipMalloc(ctypes.byref(p), 1 << 20)
lo, hi = ctypes.c_int(), ctypes.c_int()
hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
print("priority range", lo.value, hi.value)
def t(fn, label):
    s = ctypes.c_void_p()
    t0 = time.perf_counter(); rc = fn(ctypes.byref(s)); dt = (time.perf_counter() - t0) * 1e3
    print(f"{label}: {dt:.3f} ms rc={rc}")
    return s
streams = []
for i in range(4):
    streams.append(t(lambda s: hip.hipStreamCreateWithFlags(s, 1), f"plain nonblocking #{i}"))
for i in range(4):
    streams.append(t(lambda s: hip.hipStreamCreateWithPriority(s, 1, hi.value), f"high priority #{i}"))
for i in range(2):
    streams.append(t(lambda s: hip.hipStreamCreateWithPriority(s, 1, lo.value), f"low priority #{i}"))
for s in streams:
    t0 = time.perf_counter(); hip.hipStreamDestroy(s); print(f"destroy {(time.perf_counter() - t0) * 1e3:.3f} ms")
for i in range(3):
    s = t(lambda s: hip.hipStreamCreateWithPriority(s, 1, hi.value), f"high priority again #{i}")
    # first use of a stream: a memset
    t0 = time.perf_counter(); hip.hipMemsetAsync(p, 0, 1024, s); hip.hipStreamSynchronize(s); print(f"  first op {(time.perf_counter() - t0) * 1e3:.3f} ms")
    hip.hipStreamDestroy(s)
