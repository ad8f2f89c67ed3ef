"""The libm values the path takes from the HOST, as hex doubles computed with this container's glibc
(2.35): Welch divisors 4 pow(n - 1, -2) (libs/lpc/src/lpc.c:199) for every unit length of the frame sizes the tests and BASELINE
use, samples of the SIN window (lpc.c:192) and of the Cholesky pivot pow(x, -0.5) (lpc.c:421).  Data only.
Run in the build container:  python tests/golden/make_libm_golden.py"""
import ctypes as C, json, os
HERE = os.path.dirname(os.path.abspath(__file__))
m = C.CDLL("libm.so.6")
m.pow.restype = C.c_double; m.pow.argtypes = [C.c_double, C.c_double]
m.sin.restype = C.c_double; m.sin.argtypes = [C.c_double]
out = {"welch_divisor": {}, "sin_window": {}, "cholesky_pivot": {}}
for n in (10240, 9280, 4096, 3008, 2048, 2000, 1024, 1000, 680, 136, 128):
    for u in (1, 2, 4, 8, 16, 32, 64, 128):
        if n % u == 0 and n // u >= 2:
            out["welch_divisor"][str(n // u)] = (4.0 * m.pow(float(n // u - 1), -2.0)).hex()
for n in (10240, 9280, 2000, 1024, 680):
    for s in (1, 7, n // 3, n // 2, n - 2):
        out["sin_window"][f"{s}/{n}"] = m.sin((3.1415926535897932384626433832795029 * s) / (n - 1)).hex()
for k in range(40):
    x = float.fromhex("0x1.%013xp%d" % ((k * 0x9E3779B97F4A7) & 0xFFFFFFFFFFFFF, k - 20))
    out["cholesky_pivot"][x.hex()] = m.pow(x, -0.5).hex()
json.dump(out, open(os.path.join(HERE, "libm_values.json"), "w"), indent=0, sort_keys=True)
print({k: len(v) for k, v in out.items()})
