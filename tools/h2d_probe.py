"""tools/h2d_probe.py — pinned host -> device copy rate by copy size (1 MiB ... 128 MiB) on one side stream: what the link gives hipMemcpyAsync, the
yardstick of bench.py's upload-inclusive legs.  Run on the GPU box."""
import ctypes as C, time
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
def chk(e): assert e == 0, e
host = C.c_void_p(); dev = C.c_void_p(); st = C.c_void_p()
N = 256 << 20
chk(hip.hipHostMalloc(C.byref(host), C.c_size_t(N), 0)); chk(hip.hipMalloc(C.byref(dev), C.c_size_t(N))); chk(hip.hipStreamCreate(C.byref(st)))
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
for sz in (1 << 20, 2 << 20, 3 << 20, 8 << 20, 32 << 20, 128 << 20):
    n = max(4, (1 << 30) // sz)
    for rep in range(2):
        t = time.perf_counter()
        for i in range(n):
            off = (i * sz) % (N - sz)
            chk(hip.hipMemcpyAsync(dev.value + off, host.value + off, sz, 1, st))
        chk(hip.hipStreamSynchronize(st)); dt = time.perf_counter() - t
    print(f"{sz>>20:4d} MiB copies: {n*sz/dt/1e9:.1f} GB/s  ({dt/n*1e6:.1f} us each)")
