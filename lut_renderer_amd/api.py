"""`apply_lut`: the Python-side LUT-apply API with the reference's option vocabulary.

Where the reference hands a filter string to an ffmpeg child process
(/root/reference/src/lut_renderer/ffmpeg.py:195-247, :304-310, run at
task_manager.py:145-151), this applies the same chain to frames already resident in HBM:

    [scale=in_range=pc:out_range=R, format=<8-bit>]   full-range sources only
    (auto) YUV -> RGB at the negotiated depth          matrix from lut_input_matrix / colorspace
    lut3d=file=<cube>:interp=<mode>
    [zscale=dither=error_diffusion]                    option zscale_dither: the output quantisation is dithered
    [format=<pix_fmt>]                                 RGB -> YUV at the output depth

Options use the names and values of `ProcessingParams` (models.py:45-56) and `VideoInfo`
(media_info.py:25-34).  Unforced matrices fall back to BT.601, swscale's default for
untagged frames; the RGB->YUV side produces limited range, which is
what ffmpeg's auto-inserted scaler emits ahead of an encoder.
"""
from __future__ import annotations

import threading
from pathlib import Path
from typing import Dict, Optional, Sequence, Tuple

from .cube import CubeLut, read_cube, read_lut
from .params import ProcessingParams, VideoInfo, infer_bit_depth
from .plan import LutPlan, output_color_tags, resolve_lut_plan

#: FFmpeg lut3d has no "cubic"; the reference whitelists it anyway (ffmpeg.py:243) and ffmpeg
#: would reject the filtergraph.  The engine mirrors that as an error.
_ENGINE_INTERP = ("nearest", "trilinear", "tetrahedral", "pyramid", "prism")

_lut_cache: Dict[Tuple[str, float, int], CubeLut] = {}
# engines apply_lut creates for itself, kept per device tuple: a context holds a stream, the device lattice and the
# kernels' work queue, and the reference calls the path once per task, not once per frame
_engine_cache: Dict[Tuple[int, ...], object] = {}
_cache_lock = threading.Lock()      # guards the two dictionaries; each engine carries its own lock for the calls on it


def _cached_engine(devices: Tuple[int, ...]):
    from .engine import LutEngine
    from .multigpu import LutEngineGroup
    with _cache_lock:
        eng = _engine_cache.get(devices)
        if eng is None:
            eng = LutEngine(devices[0]) if len(devices) == 1 else LutEngineGroup(devices)
            _engine_cache[devices] = eng
        return eng


def close_cached_engines() -> None:
    """Destroy the contexts `apply_lut` keeps between calls (also registered with atexit)."""
    with _cache_lock:
        engines = list(_engine_cache.values())
        _engine_cache.clear()
    for eng in engines:
        try:
            with eng._lock:
                eng.close()
        except Exception:
            pass


import atexit  # noqa: E402
atexit.register(close_cached_engines)


def _cached_cube(path: Path) -> CubeLut:
    st = path.stat()
    key = (str(path), st.st_mtime, st.st_size)
    with _cache_lock:
        lut = _lut_cache.get(key)
    if lut is None:
        lut = read_lut(path)                   # parsed outside the lock: two tasks may parse two files at once
        with _cache_lock:
            if len(_lut_cache) >= 16:          # up to 16 concurrent tasks (task_manager.py:229-235), each with its own LUT
                _lut_cache.clear()
            lut = _lut_cache.setdefault(key, lut)
    return lut


def engine_call_for(plan: LutPlan, pix_fmt: str, out_pix_fmt: Optional[str] = None) -> dict:
    """Translate a LutPlan into keyword arguments of LutEngine.apply_yuv."""
    from .engine import parse_pix_fmt
    if plan.interp not in _ENGINE_INTERP:
        raise ValueError(f"lut3d has no interpolation mode '{plan.interp}'")
    src = parse_pix_fmt(pix_fmt)
    if src.family != "yuv":
        raise ValueError("apply_lut takes planar YUV frames; use LutEngine.apply_rgb for gbrp planes")
    matrix = plan.matrix or "smpte170m"
    kw = dict(pix_fmt=pix_fmt.replace("yuvj", "yuv"), interp=plan.interp, matrix_in=matrix, matrix_out=matrix,
              range_out="tv")
    if plan.prologue:
        # scale=in_range=pc:out_range=R , format=yuv4xxp (8 bit): the LUT then runs at 8 bit
        kw.update(range_src="pc", range_in=plan.prologue_out_range, lut_depth=8)
        default_out = plan.intermediate_pix_fmt
    else:
        kw.update(range_src="tv", range_in="tv", lut_depth=src.depth)
        default_out = kw["pix_fmt"]
    kw["out_pix_fmt"] = out_pix_fmt or default_out
    return kw


def apply_lut(planes: Sequence, *, cube, interp: str = "tetrahedral", pix_fmt: str, width: Optional[int] = None,
              height: Optional[int] = None, input_matrix: str = "auto", colorspace: Optional[str] = None,
              color_range: Optional[str] = None, output_tags: str = "bt709", out_pix_fmt: Optional[str] = None,
              zscale_dither: str = "none", out: Optional[Sequence] = None, engine=None,
              devices: Sequence[int] = (0,), precision: str = "strict"):
    """Apply `cube` to planar YUV frames on the GPU.  `planes` = (Y, Cb, Cr) torch tensors on the
    engine's device, each [H,W] or [F,H,W].  Returns (planes_out, tags) where `tags` is the colour
    metadata the reference would write for this policy (None = inherit / none).

    `engine` may be a LutEngine (or LutEngineGroup) that already holds the lattice (then `cube` may be
    None).  Otherwise `devices` names the GPUs: one device -> one context; several -> a LutEngineGroup that
    splits the rows of every frame over them inside this one process (lattice copied GPU to GPU, one launch
    per device, no host wait between launches).  Contexts made here are kept for the next call
    (`close_cached_engines`) and are safe to share between the threads of a task pool: the engine's lock is held from
    the lattice upload to the launch.

    `precision` is an ENGINE setting, not one of the reference's (models.py:45-56 has no such field): "strict" (default)
    is the bit-exact restatement of FFmpeg's scalar C in fp32; "fast" allows the tolerance-bounded kernels whose lattice
    is fp16 (<= 1 code from strict at 8 and 10 bit) where they exist and silently runs strict elsewhere
    (`engine.last_kernel` ends in `,fast` when they ran)."""
    devices = tuple(int(d) for d in devices)
    if not devices:
        raise ValueError("devices must name at least one GPU")
    if width is not None and planes[0].shape[-1] != width or height is not None and planes[0].shape[-2] != height:
        raise ValueError("plane shape does not match width/height")
    params = ProcessingParams(lut_interp=interp, lut_input_matrix=input_matrix, lut_output_tags=output_tags,
                              zscale_dither=zscale_dither)
    info = VideoInfo(width=width, height=height, pix_fmt=pix_fmt, bit_depth=infer_bit_depth(pix_fmt),
                     colorspace=colorspace, color_range=color_range)
    # (the plan only carries the path into the filter string / notes; a parsed CubeLut or an engine that already holds the
    # lattice has none)
    plan = resolve_lut_plan(params, cube if isinstance(cube, (str, Path)) else "engine.cube", info)
    kw = engine_call_for(plan, pix_fmt, out_pix_fmt)
    # ffmpeg.py:305-307: any value other than "error_diffusion" leaves the chain without a dither filter
    kw["dither"] = "error_diffusion" if getattr(params, "zscale_dither", "none") == "error_diffusion" else "none"
    if precision not in ("strict", "fast"):
        raise ValueError(f"unknown precision '{precision}' (strict | fast)")
    own = engine is None
    eng = engine if engine is not None else _cached_engine(devices)
    lut = None
    if cube is not None:
        lut = cube if isinstance(cube, CubeLut) else _cached_cube(Path(cube))
    with eng._lock:                 # upload, precision and launch of ONE task: no other thread's call gets in between
        if lut is not None and eng._applied_lut is not lut:      # same parsed LUT as last time: the device copy stands
            eng.set_lut(lut)                                     # (set_lut itself clears the marker)
            eng._applied_lut = lut
        if eng.precision != precision:
            eng.set_precision(precision)
        result = eng.apply_yuv(planes, out, **kw)
        if own:
            eng.sync()
    return result, output_color_tags(plan.output_policy)
