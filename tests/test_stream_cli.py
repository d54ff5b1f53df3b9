"""Host-queue pipeline (BASELINE config 5) and the engine CLI's process contract
(/root/reference/src/lut_renderer/task_manager.py:14-15, :134-190)."""
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from lut_renderer_amd import cube, frames

ROOT = Path(__file__).resolve().parent.parent
_DURATION_RE = re.compile(r"Duration: (?P<h>\d+):(?P<m>\d+):(?P<s>\d+(?:\.\d+)?)")     # task_manager.py:14
_TIME_RE = re.compile(r"time=(?P<h>\d+):(?P<m>\d+):(?P<s>\d+(?:\.\d+)?)")             # task_manager.py:15


def test_frame_layout_views():
    import torch
    from lut_renderer_amd.engine import parse_pix_fmt
    from lut_renderer_amd.stream import FrameLayout
    lay = FrameLayout(parse_pix_fmt("yuv420p10le"), 64, 36)
    assert lay.plane_bytes == [64 * 36 * 2, 32 * 18 * 2, 32 * 18 * 2] and lay.frame_bytes == 64 * 36 * 3
    buf = torch.arange(2 * lay.frame_bytes, dtype=torch.int64).to(torch.uint8)
    y, cb, cr = lay.plane_views(buf, 2)
    assert y.shape == (2, 36, 64) and cb.shape == (2, 18, 32)
    assert y.stride() == (lay.frame_bytes // 2, 64, 1)
    assert cb.storage_offset() == 64 * 36 and cr.storage_offset() == 64 * 36 + 32 * 18


def test_cli_reports_errors_with_exit_code(tmp_path):
    r = subprocess.run([sys.executable, "-m", "lut_renderer_amd.cli", "-i", str(tmp_path / "none.yuv"), "-o",
                        str(tmp_path / "o.yuv"), "--size", "64x36", "--pix-fmt", "yuv420p", "--cube",
                        str(tmp_path / "none.cube")], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 1 and r.stdout.startswith("Error:")


@pytest.mark.gpu
def test_pipeline_matches_oracle(engine, orc, cube_dir):
    from lut_renderer_amd.stream import HostPipeline
    lut = cube.read_cube(cube_dir / "log709_33.cube")
    engine.set_lut(lut)
    w, h, n = 128, 72, 11                      # 11 frames, batch 4 -> a ragged last batch, ring of 2 wraps
    src = [frames.natural_yuv(w, h, 10, 1, 1, k=i) for i in range(n)]
    raw = b"".join(p.tobytes() for f in src for p in f)
    pipe = HostPipeline(engine, "yuv420p10le", w, h, batch=4, slots=2, interp="tetrahedral")
    fb = pipe.fin.frame_bytes
    pos, out = [0], []

    def fill(buf, m):
        k = min(m, n - pos[0])
        buf[: k * fb] = np.frombuffer(raw, dtype=np.uint8, count=k * fb, offset=pos[0] * fb)
        pos[0] += k
        return k

    done = pipe.run(fill, lambda b, k: out.append(bytes(b)), total_frames=n)
    assert done == n
    got = np.frombuffer(b"".join(out), dtype=np.uint16)
    k = orc.yuv_constants(din=10)
    want = np.concatenate([p.ravel() for f in src
                           for p in orc.apply_yuv(lut.table, lut.scale, "tetrahedral", k, 10, 10, 10, 1, 1, f)])
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_cli_process_contract(orc, cube_dir, tmp_path):
    """Popen'd like TaskRunner does: Duration once, time= lines, exit 0, output == oracle.
    Full-range 8-bit source: the plan inserts the pc->tv prologue (SURVEY.md Appendix D case A)."""
    w, h, n = 64, 36, 5
    src = [frames.uniform_yuv(w, h, 8, 1, 1, k=i, full_range=True) for i in range(n)]
    (tmp_path / "in.yuv").write_bytes(b"".join(p.tobytes() for f in src for p in f))
    cmd = [sys.executable, "-m", "lut_renderer_amd.cli", "-y", "-i", str(tmp_path / "in.yuv"), "-o",
           str(tmp_path / "out.yuv"), "--size", f"{w}x{h}", "--pix-fmt", "yuvj420p", "--cube",
           str(cube_dir / "log709_33.cube"), "--interp", "trilinear", "--colorspace", "bt709", "--color-range", "pc",
           "--fps", "25", "--batch", "2"]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, bufsize=1, cwd=ROOT)
    lines = [line for line in proc.stdout]
    assert proc.wait(timeout=120) == 0, "".join(lines)
    durations = [m for m in map(_DURATION_RE.search, lines) if m]
    times = [m for m in map(_TIME_RE.search, lines) if m]
    assert len(durations) == 1 and float(durations[0].group("s")) == pytest.approx(n / 25.0)
    assert len(times) >= 2 and float(times[-1].group("s")) == pytest.approx(n / 25.0)
    lut = cube.read_cube(cube_dir / "log709_33.cube")
    k = orc.yuv_constants("bt709", "tv", "bt709", "tv", 8, 8, 8, 4, prologue=True)
    want = b"".join(p.tobytes() for f in src
                    for p in orc.apply_yuv(lut.table, lut.scale, "trilinear", k, 8, 8, 8, 1, 1, f))
    assert (tmp_path / "out.yuv").read_bytes() == want
