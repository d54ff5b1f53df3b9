"""AddressSanitizer + UBSan over the code that reads untrusted files and over the oracle (CPU box only --
GPU sanitizers are not available on this pool).  `make -C oracle asan` builds the oracle and a HIP-free build of the
product's two host parsers (csrc/cube_parse.cpp, csrc/lut_formats.cpp) with gcc -fsanitize=address,undefined; a child
python with libasan preloaded feeds them valid, truncated, garbage and huge-line files and runs every oracle mode.
Any sanitizer report aborts the child, which fails the test."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent

CHILD = r'''
import ctypes as C, os, sys, random
import numpy as np
root, tmp = sys.argv[1], sys.argv[2]
par = C.CDLL(os.path.join(root, "oracle/_build/liblutr_parsers_asan.so"))
orc = C.CDLL(os.path.join(root, "oracle/_build/liblut3d_oracle_asan.so"))
par.lutr_lut_parse.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int), C.POINTER(C.c_float)]
par.lutr_cube_parse.argtypes = par.lutr_lut_parse.argtypes
par.lutr_cube_free.argtypes = [C.POINTER(C.c_float)]
par.lutr_cube_free.restype = None

class OrcLut(C.Structure):
    _fields_ = [("n", C.c_int), ("scale", C.c_float * 3), ("rgb", C.POINTER(C.c_float)),
                ("pre_size", C.c_int), ("pre_min", C.c_float * 3), ("pre_scale", C.c_float * 3), ("prelut", C.POINTER(C.c_float))]
par.lutr_lut_parse_ex.argtypes = par.lutr_lut_parse.argtypes + [C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int),
                                                               C.POINTER(C.c_float), C.POINTER(C.c_float)]
orc.orc_lut_file_parse.argtypes = [C.c_char_p, C.POINTER(OrcLut)]
orc.orc_lut_free.argtypes = [C.POINTER(OrcLut)]
orc.orc_lut_free.restype = None

def both(path):
    rgb, n, sc = C.POINTER(C.c_float)(), C.c_int(), (C.c_float * 3)()
    pre, psz, pmin, psc = C.POINTER(C.c_float)(), C.c_int(), (C.c_float * 3)(), (C.c_float * 3)()
    rc = par.lutr_lut_parse_ex(path.encode(), C.byref(rgb), C.byref(n), sc, C.byref(pre), C.byref(psz), pmin, psc)
    if rc == 0:
        a = np.ctypeslib.as_array(rgb, shape=(n.value ** 3 * 3,)).copy()      # touch every float the parser returned
        assert np.isfinite(a).all()
        par.lutr_cube_free(rgb)
        if psz.value:
            assert np.isfinite(np.ctypeslib.as_array(pre, shape=(3 * psz.value,)).copy()).all()
            par.lutr_cube_free(pre)
    lut = OrcLut()
    rc2 = orc.orc_lut_file_parse(path.encode(), C.byref(lut))
    if rc2 == 0:
        np.ctypeslib.as_array(lut.rgb, shape=(lut.n ** 3 * 3,)).copy()
        if lut.pre_size:
            np.ctypeslib.as_array(lut.prelut, shape=(3 * lut.pre_size,)).copy()
        orc.orc_lut_free(C.byref(lut))
    return rc, rc2

def w(name, data):
    p = os.path.join(tmp, name)
    with open(p, "wb") as f:
        f.write(data if isinstance(data, bytes) else data.encode())
    return p

rng = random.Random(7)
tri = lambda: "%d %d %d\n" % (rng.randrange(4096), rng.randrange(4096), rng.randrange(4096))
ent = lambda n: "".join("%.6f %.6f %.6f\n" % (rng.random(), rng.random(), rng.random()) for _ in range(n))
good = {
    "a.cube": "TITLE \"x\"\n# c\nLUT_3D_SIZE 4\nDOMAIN_MIN 0 0 0\nDOMAIN_MAX 1 1 1\n" + ent(64),
    "a.dat": "3DLUTSIZE 4\n" + ent(64),
    "a.3dl": "# h\n" + " ".join(str(min(i * 64, 1023)) for i in range(17)) + "\n" + "".join(tri() for _ in range(17 ** 3)),
    "a.m3d": "name x\nin 27\nout 4096\nformat lut\nvalues\tblue\tgreen\tred\n" + "".join(tri() for _ in range(27)),
    "a.csp": "CSPLUTV100\n3D\n\n2\n0.0 1.0\n0.0 1.0\n2\n0.0 1.0\n0.0 1.0\n2\n0.0 1.0\n0.0 1.0\n\n3 3 3\n" + ent(27),
    "b.csp": "CSPLUTV100\n3D\n\n" + "5\n0.0 0.1 0.4\n0.7 1.0\n0.0 0.3 0.6 0.8 1.0\n" * 3 + "\n3 3 3\n" + ent(27),      # pre-LUTs
}
n_ok = 0
for name, text in good.items():
    p = w(name, text)
    rc, rc2 = both(p)
    n_ok += (rc == 0) + (rc2 == 0)
    stem, ext = name.split(".")
    data = text.encode()
    # truncated at many points, including mid-number and mid-header
    for cut in sorted({0, 1, 5, 11, 12, 13, 20, len(data) // 3, len(data) // 2, len(data) - 7, len(data) - 1}):
        both(w(f"t{cut}.{ext}", data[:max(0, cut)]))
    # a 70 kB line in the header, in the table, and unterminated at EOF (fgets returns it in 512-byte pieces)
    long = "9" * 70000
    both(w(f"l1.{ext}", long + "\n" + text))
    both(w(f"l2.{ext}", text.replace("\n", " " + long + "\n", 3)))
    both(w(f"l3.{ext}", text + "0.1 0.2 " + long))
    # binary junk, NULs, huge / negative / non-numeric sizes, nan and inf entries
    both(w(f"j1.{ext}", bytes(rng.randrange(256) for _ in range(4096))))
    both(w(f"j2.{ext}", b"\0" * 2048 + data))
    for size in ("0", "1", "257", "-3", "99999999999999999999", "4x", ""):
        both(w(f"s.{ext}", text.replace("LUT_3D_SIZE 4", "LUT_3D_SIZE " + size).replace("3DLUTSIZE 4", "3DLUTSIZE " + size)
               .replace("in 27", "in " + size).replace("3 3 3", f"{size} 3 3")))
    both(w(f"n.{ext}", text.replace("\n", "\nnan inf -inf\n", 1)))
both(w("noext", good["a.cube"]))
both(os.path.join(tmp, "does-not-exist.cube"))
assert n_ok == 2 * len(good), n_ok

# every oracle mode on small frames (threads included), odd sizes, both containers
P3, S3 = C.c_void_p * 3, C.c_ssize_t * 3
orc.orc_apply_planar_rgb.argtypes = [C.POINTER(OrcLut), C.c_int, C.c_int, C.c_int, C.c_int, P3, S3, P3, S3, C.c_int]
tab = np.random.default_rng(3).uniform(-0.1, 1.1, size=(5, 5, 5, 3)).astype(np.float32)
lut = OrcLut(); lut.n = 5
for i in range(3): lut.scale[i] = 1.0
lut.rgb = tab.ctypes.data_as(C.POINTER(C.c_float))
for depth, dt in ((8, np.uint8), (10, np.uint16), (16, np.uint16)):
    for (h, wd) in ((1, 1), (7, 13), (32, 64)):
        src = [np.random.default_rng(i).integers(0, 1 << depth, size=(h, wd)).astype(dt) for i in range(3)]
        dst = [np.zeros_like(s) for s in src]
        for mode in range(5):
            rc = orc.orc_apply_planar_rgb(C.byref(lut), depth, mode, wd, h, P3(*[s.ctypes.data for s in src]),
                                          S3(*[s.strides[0] for s in src]), P3(*[d.ctypes.data for d in dst]),
                                          S3(*[d.strides[0] for d in dst]), 3)
            assert rc == 0
print("sanitizers ok")
'''


def test_parsers_and_oracle_under_asan_ubsan(tmp_path):
    made = subprocess.run(["make", "-C", str(ROOT / "oracle"), "asan"], capture_output=True, text=True)
    assert made.returncode == 0, made.stdout + made.stderr
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("gcc has no libasan.so here")
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    res = subprocess.run([sys.executable, "-c", CHILD, str(ROOT), str(tmp_path)], env=env, capture_output=True, text=True,
                         timeout=600)
    assert res.returncode == 0 and "sanitizers ok" in res.stdout, (res.stdout[-2000:], res.stderr[-6000:])
