import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rssync_amd
lib = rssync_amd.load_library()
lib.rship_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
lib.rship_debug_select.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
lib.rship_last_error.restype = C.c_char_p; lib.rship_last_error.argtypes = [C.c_void_p]
ctx = C.c_void_p()
assert lib.rship_create(C.byref(ctx), -1) == 0
rng = np.random.default_rng(0)
def run(vals, kq, upper=None):
    P, n = vals.shape
    out = np.zeros((P, 2), dtype=np.uint32)
    v = np.ascontiguousarray(vals, dtype=np.float32)
    u = np.ascontiguousarray(upper, dtype=np.float32) if upper is not None else None
    rc = lib.rship_debug_select(ctx, v.ctypes.data, P, n, kq, u.ctypes.data if u is not None else None, out.ctypes.data)
    assert rc == 0, lib.rship_last_error(ctx)
    return out
for n, kq in [(2048, 512), (256, 64), (100, 25), (3, 0), (5, 1), (2048, 0), (2048, 2047), (17, 4)]:
    for dist in ("sq", "exp", "ties", "const"):
        P = 400
        if dist == "sq": vals = (rng.normal(size=(P, n)) * 1e-3) ** 2
        elif dist == "exp": vals = np.exp(rng.uniform(-40, 5, size=(P, n)))
        elif dist == "ties": vals = rng.integers(0, 7, size=(P, n)).astype(np.float32) * 0.125
        else: vals = np.full((P, n), 0.25)
        vals = vals.astype(np.float32)
        want = np.sort(vals, axis=1)[:, kq].view(np.uint32)
        got = run(vals, kq)
        bad = (got[:, 0] != want).sum()
        # with a bound: just above / at / below the true quantile
        srt = np.sort(vals, axis=1)
        kth = srt[:, kq]
        for name, up in (("above", np.nextafter(kth, np.float32(np.inf)) * np.float32(1.5) + np.float32(1e-30)), ("at", kth), ("next", np.nextafter(kth, np.float32(np.inf)))):
            g2 = run(vals, kq, up)
            cnt = (vals < up[:, None]).sum(1)
            exp = np.where(cnt > kq, want, 0xffffffff)
            b2 = (g2[:, 0] != exp).sum() + (g2[:, 1] != cnt).sum()
            if b2: print(f"  n={n} kq={kq} {dist} bound={name}: {b2} wrong")
        print(f"n={n} kq={kq} {dist}: {bad}/{P} wrong (no bound)")
