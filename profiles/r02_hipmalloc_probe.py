# hipMalloc / hipFree cost on the box, by size (upload of the benchmark database spends 0.25 of its 0.63 s in eight 400 MB hipMallocs):
#   python profiles/r02_hipmalloc_probe.py
import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipSetDevice(0)
p = C.c_void_p()
hip.hipMalloc(C.byref(p), 1 << 20); hip.hipFree(p)
for mb in (4, 64, 400, 400, 400, 1600, 3200, 400):
    t0 = time.perf_counter(); rc = hip.hipMalloc(C.byref(p), mb << 20); t1 = time.perf_counter()
    hip.hipFree(p); t2 = time.perf_counter()
    print("hipMalloc %5d MB: %7.2f ms   hipFree %7.2f ms  (rc %d)" % (mb, (t1 - t0) * 1e3, (t2 - t1) * 1e3, rc))
ps = []
t0 = time.perf_counter()
for _ in range(8):
    q = C.c_void_p(); hip.hipMalloc(C.byref(q), 400 << 20); ps.append(q)
t1 = time.perf_counter()
print("8 x 400 MB held together: %.2f ms" % ((t1 - t0) * 1e3))
for q in ps: hip.hipFree(q)
# pinned host memory (candidate for the upload's staging buffers)
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipHostFree.argtypes = [C.c_void_p]
for mb in (64, 1024, 3584, 3584):
    t0 = time.perf_counter(); rc = hip.hipHostMalloc(C.byref(p), mb << 20, 0); t1 = time.perf_counter()
    C.memset(p, 1, mb << 20); t2 = time.perf_counter()
    hip.hipHostFree(p); t3 = time.perf_counter()
    print("hipHostMalloc %5d MB: %7.1f ms   first touch (1 thread) %7.1f ms   hipHostFree %7.1f ms  (rc %d)" % (mb, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, rc))
