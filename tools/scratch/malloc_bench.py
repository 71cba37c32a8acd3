import ctypes, time, threading
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipSetDevice(0)
p = ctypes.c_void_p()
hip.hipMalloc(ctypes.byref(p), 1 << 20); hip.hipFree(p)
def seq(gb, count):
    ps = []
    ts = []
    for i in range(count):
        q = ctypes.c_void_p()
        t0 = time.time(); rc = hip.hipMalloc(ctypes.byref(q), ctypes.c_size_t(int(gb * (1 << 30)))); ts.append((time.time() - t0) * 1e3)
        ps.append(q)
    t0 = time.time()
    for q in ps: hip.hipFree(q)
    print("%d x %.2f GB: malloc ms %s ; free all %.1f ms" % (count, gb, " ".join("%.1f" % t for t in ts), (time.time() - t0) * 1e3))
seq(3.0, 20)
seq(3.9, 10)
seq(4.1, 6)
seq(6.0, 6)
seq(3.0, 20)
def worker():
    hip.hipSetDevice(0)
    seq(3.0, 8)
th = [threading.Thread(target=worker) for _ in range(4)]
t0 = time.time()
for t in th: t.start()
for t in th: t.join()
print("4 threads x 8 x 3 GB: %.1f ms" % ((time.time() - t0) * 1e3))
