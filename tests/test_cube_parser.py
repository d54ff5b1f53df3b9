"""liblutr's .cube reader (csrc/cube_parse.cpp) against the oracle's reader and the NumPy twin.

Three separately written restatements of FFmpeg's parse_cube -- the file the
reference names in lut3d=file=... at /root/reference/src/lut_renderer/ffmpeg.py:246.
No GPU needed: parsing is host code.
"""
import numpy as np
import pytest

from lut_renderer_amd import _native, cube
from lut_renderer_amd._native import LutrError
from oracle import lut3d_numpy as npo


def _write(tmp_path, name, text):
    p = tmp_path / name
    p.write_text(text)
    return p


def _entries(n, fn=lambda i: (i * 0.001, i * 0.002, i * 0.003)):
    return "".join("%.6f %.6f %.6f\n" % fn(i) for i in range(n ** 3))


def _both(orc, path):
    lut = cube.read_cube(path)
    n, sc, tab = orc.parse_cube(path)
    assert lut.n == n
    assert np.array_equal(lut.scale, sc)
    assert np.array_equal(lut.table, tab)
    n3, sc3, tab3 = npo.parse_cube_text(open(path).read())
    assert n3 == n and np.array_equal(tab3, tab) and np.array_equal(sc3, sc)
    return lut


def test_red_varies_fastest(orc, tmp_path):
    """A.6 #7: file entry 1 lands at lut[r=1,g=0,b=0]; entry n at g=1; entry n*n at b=1."""
    n = 3
    p = _write(tmp_path, "o.cube", f"LUT_3D_SIZE {n}\n" + _entries(n, lambda i: (float(i), 0.0, 0.5)))
    lut = _both(orc, p)
    assert lut.table[1, 0, 0, 0] == 1.0
    assert lut.table[0, 1, 0, 0] == float(n)
    assert lut.table[0, 0, 1, 0] == float(n * n)
    assert lut.table[2, 2, 2, 0] == float(n ** 3 - 1)


def test_comments_blank_title_and_domain_positions(orc, tmp_path):
    body = _entries(2)
    lines = body.splitlines(keepends=True)
    text = ("# header comment\nTITLE \"before size\"\nDOMAIN_MAX 4 4 4\nLUT_3D_INPUT_RANGE 0 1\n"
            "LUT_3D_SIZE 2\n\n   # indented comment\nTITLE \"after size\"\nDOMAIN_MIN 0 0 0\n"
            + lines[0] + "\n" + lines[1] + "DOMAIN_MAX 2.0 4.0 8.0\n" + "".join(lines[2:]) + "trailing junk ignored\n")
    lut = _both(orc, _write(tmp_path, "c.cube", text))
    # DOMAIN_MAX before LUT_3D_SIZE is ignored (quirk); the one inside the table counts
    assert np.allclose(lut.scale, [0.5, 0.25, 0.125])


def test_domain_scale_clip_and_min_not_subtracted(orc, tmp_path):
    text = "LUT_3D_SIZE 2\nDOMAIN_MIN 0.25 0 0\nDOMAIN_MAX 0.75 1 3\n" + _entries(2)
    lut = _both(orc, _write(tmp_path, "d.cube", text))
    assert np.allclose(lut.scale, [1.0, 1.0, 1.0 / 3.0])    # 1/0.5 = 2 clips to 1


@pytest.mark.parametrize("size,code", [(1, _native.EINVAL), (257, _native.EINVAL), (0, _native.EINVAL)])
def test_bad_sizes(orc, tmp_path, size, code):
    p = _write(tmp_path, "s.cube", f"LUT_3D_SIZE {size}\n" + "0 0 0\n" * 8)
    with pytest.raises(LutrError) as e:
        cube.read_cube(p)
    assert e.value.code == code
    from oracle.binding import OracleError
    with pytest.raises(OracleError) as eo:
        orc.parse_cube(p)
    assert eo.value.code == code


def test_max_size_256_parses(orc, tmp_path):
    """LUT_3D_SIZE 256 is the largest legal lattice (16.7M entries); check size handling on a
    sparse file by truncating: the error must be 'unexpected EOF', not a size rejection."""
    p = _write(tmp_path, "big.cube", "LUT_3D_SIZE 256\n" + "0.5 0.5 0.5\n" * 100)
    with pytest.raises(LutrError) as e:
        cube.read_cube(p)
    assert e.value.code == _native.EILSEQ and "EOF" in e.value.message


def test_truncated_and_garbage(orc, tmp_path):
    from oracle.binding import OracleError
    cases = {
        "trunc.cube": "LUT_3D_SIZE 2\n" + "0 0 0\n" * 7,
        "garbage.cube": "LUT_3D_SIZE 2\n" + "0 0 0\n" * 3 + "zero one two\n" + "0 0 0\n" * 4,
        "two_numbers.cube": "LUT_3D_SIZE 2\n" + "0 0 0\n" * 3 + "0.1 0.2\n" + "0 0 0\n" * 4,
        "nosize.cube": "TITLE x\n" + "0 0 0\n" * 8,
        "empty.cube": "",
        "bad_domain.cube": "LUT_3D_SIZE 2\nDOMAIN_MID 1 1 1\n" + "0 0 0\n" * 8,
    }
    for name, text in cases.items():
        p = _write(tmp_path, name, text)
        with pytest.raises(LutrError) as e:
            cube.read_cube(p)
        assert e.value.code == _native.EILSEQ, name
        with pytest.raises(OracleError) as eo:
            orc.parse_cube(p)
        assert eo.value.code == _native.EILSEQ, name


def test_extension_and_missing_file(orc, tmp_path):
    from oracle.binding import OracleError
    good = "LUT_3D_SIZE 2\n" + _entries(2)
    _both(orc, _write(tmp_path, "UPPER.CUBE", good))             # extension match is case-insensitive
    for name in ("lut.3dl", "noext"):
        p = _write(tmp_path, name, good)
        with pytest.raises(LutrError) as e:
            cube.read_cube(p)
        assert e.value.code == _native.EINVAL
        with pytest.raises(OracleError) as eo:
            orc.parse_cube(p)
        assert eo.value.code == _native.EINVAL
    with pytest.raises(LutrError) as e:
        cube.read_cube(tmp_path / "missing.cube")
    assert e.value.code == _native.ENOENT


def test_number_formats(orc, tmp_path):
    text = "LUT_3D_SIZE 2\n" + "1e-3 .5 1.\n  0.25\t0.5   0.75  # trailing comment\n-0.5 +1.5 1E+0\n" + "0 0 0\n" * 5
    lut = _both(orc, _write(tmp_path, "f.cube", text))
    assert np.allclose(lut.table[0, 0, 0], [1e-3, 0.5, 1.0])
    assert np.allclose(lut.table[1, 0, 0], [0.25, 0.5, 0.75])
    assert np.allclose(lut.table[0, 1, 0], [-0.5, 1.5, 1.0])


def test_crlf_line_endings(orc, tmp_path):
    p = tmp_path / "crlf.cube"
    p.write_bytes(("TITLE \"x\"\r\nLUT_3D_SIZE 2\r\n\r\n" + _entries(2).replace("\n", "\r\n")).encode())
    lut = cube.read_cube(p)
    n, sc, tab = orc.parse_cube(p)
    assert lut.n == n == 2 and np.array_equal(lut.table, tab)


def test_non_finite_values_are_rejected_by_the_product(orc, tmp_path):
    """Documented deviation (csrc/cube_parse.cpp): sscanf accepts nan/inf, the engine does not."""
    p = _write(tmp_path, "nan.cube", "LUT_3D_SIZE 2\n" + "0 0 0\n" * 3 + "nan 0 0\n" + "0 0 0\n" * 4)
    with pytest.raises(LutrError) as e:
        cube.read_cube(p)
    assert e.value.code == _native.EILSEQ and "non-finite" in e.value.message
    n, _, tab = orc.parse_cube(p)           # the oracle keeps FFmpeg's behaviour
    assert n == 2 and np.isnan(tab[1, 1, 0, 0])


def test_write_read_round_trip(orc, tmp_path):
    for name, n in (("identity", 17), ("log709", 33)):
        lat = cube.generate(name, n)
        lut = _both(orc, cube.write_cube(tmp_path / f"{name}.cube", lat, title=name))
        assert np.abs(lut.table - lat).max() < 1e-6            # %.6f text
    assert np.array_equal(cube.read_cube(tmp_path / "identity.cube").scale, np.ones(3, dtype=np.float32))
