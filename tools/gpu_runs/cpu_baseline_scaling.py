"""cpu_baseline harness (oracle/gl_oracle.c glo_fft_bench) by thread count on this host; with OLD=<path to an older
libgl_oracle.so> the same for that build. Test infrastructure, not product."""
import ctypes, json, os, sys
import numpy as np
def load(path):
    L = ctypes.CDLL(path)
    L.glo_fft_bench.restype = ctypes.c_double
    L.glo_fft_bench.argtypes = [ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
    L.glo_hardware_threads.restype = ctypes.c_int
    return L
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
libs = {"new": load(os.path.join(root, "oracle", "libgl_oracle.so"))}
if os.environ.get("OLD"):
    libs["old"] = load(os.environ["OLD"])
hw = libs["new"].glo_hardware_threads()
n = 1 << 20
out = {"hardware_threads": hw}
for tag, L in libs.items():
    rates = {}
    for t in sorted({1, 8, 32, 64, 128, hw // 2, hw}):
        if t > hw: continue
        bad = ctypes.c_uint64(0)
        best = 0.0
        for rep in range(2):
            dt = L.glo_fft_bench(n, t, 2, 0x706C6F6E6B7932 + t + rep, ctypes.byref(bad))
            assert bad.value == 0
            best = max(best, 2 * t * 2 / dt)
        rates[t] = round(best, 1)
    out[tag] = rates
print(json.dumps(out))
