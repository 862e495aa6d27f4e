import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
p = C.c_void_p()
hip.hipSetDevice(0)
hip.hipMalloc(C.byref(p), C.c_size_t(1<<20)); hip.hipFree(p)
for sz in (1<<20, 16<<20, 243<<20, 1<<30):
    ts=[]
    for it in range(6):
        t0=time.perf_counter(); hip.hipMalloc(C.byref(p), C.c_size_t(sz)); t1=time.perf_counter(); hip.hipFree(p); t2=time.perf_counter()
        ts.append(((t1-t0)*1e6,(t2-t1)*1e6))
    print(sz>>20, "MB malloc/free us:", ["%.0f/%.0f"%t for t in ts])
