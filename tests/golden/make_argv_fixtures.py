#!/usr/bin/env python3
"""Regenerates tests/golden/argv_cases.json by IMPORTING the reference's command builder
(/root/reference/src/lut_renderer/ffmpeg.py, stdlib only) in this container.  The reference
cannot travel to the GPU box; the JSON (inputs + the argv / notes / errors it produced) can.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_argv_fixtures.py
"""
import dataclasses
import json
import sys
from pathlib import Path

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/src")
from lut_renderer.ffmpeg import _escape_filter_path, build_command, build_pipeline  # noqa: E402
from lut_renderer.media_info import VideoInfo  # noqa: E402
from lut_renderer.models import ProcessingParams, Task  # noqa: E402

INFOS = {
    "none": None,
    "pc8": dict(bit_depth=8, pix_fmt="yuvj420p", color_range="pc", colorspace="bt709"),
    "pc10": dict(bit_depth=10, pix_fmt="yuv422p10le", color_range="pc", colorspace="bt2020nc"),
    "pc444": dict(bit_depth=8, pix_fmt="yuvj444p", color_range="pc", colorspace="smpte170m"),
    "tv10": dict(bit_depth=10, pix_fmt="yuv420p10le", color_range="tv", colorspace="bt2020nc", fps=24.0),
    "tv8": dict(bit_depth=8, pix_fmt="yuv420p", color_range="tv", colorspace="bt709", fps=29.97, avg_fps=29.97,
                r_fps=30.0, is_vfr=True, color_primaries="bt709", color_trc="bt709", bitrate="60000k"),
    "fcc": dict(bit_depth=8, pix_fmt="yuv420p", colorspace="fcc"),
    "untagged10": dict(bit_depth=10, pix_fmt="yuv420p10le"),
}

PARAM_SETS = {
    "x265": dict(video_codec="libx265"),
    "x264": dict(video_codec="libx264"),
    "x265_inherit": dict(video_codec="libx265", lut_output_tags="inherit"),
    "x265_none": dict(video_codec="libx265", lut_output_tags="none"),
    "x265_weirdtags": dict(video_codec="libx265", lut_output_tags="P3"),
    "x265_matrix_none": dict(video_codec="libx265", lut_input_matrix="none"),
    "x265_matrix_709": dict(video_codec="libx265", lut_input_matrix="bt709"),
    "x265_matrix_601": dict(video_codec="libx265", lut_input_matrix="SMPTE170M"),
    "x265_bogus_interp": dict(video_codec="libx265", lut_interp="bogus"),
    "x265_cubic": dict(video_codec="libx265", lut_interp="cubic"),
    "x265_trilinear": dict(video_codec="libx265", lut_interp="trilinear"),
    "x264_8bit_dither": dict(video_codec="libx264", bit_depth_policy="force_8bit", zscale_dither="error_diffusion"),
    "prores": dict(video_codec="prores_ks"),
    "vt_hi": dict(video_codec="h264_videotoolbox", bitrate="80M"),
    "x264_rates": dict(video_codec="libx264", bitrate="7.5M", crf="18", preset="slow", tune="film", gop="48",
                       profile="high", level="4.1", threads="8", resolution="1920x1080", faststart=True,
                       audio_bitrate="192k", sample_rate="48000", channels="2"),
    "x264_fps": dict(video_codec="libx264", fps="30000/1001"),
    "x264_nocfr": dict(video_codec="libx264", force_cfr=False),
    "x264_p_trilinear_709_none_8bit": dict(video_codec="libx264", lut_interp="trilinear", lut_input_matrix="bt709",
                                           lut_output_tags="none", bit_depth_policy="force_8bit"),
    "copy": dict(video_codec="copy"),
    "pixfmt_forced": dict(video_codec="libx265", pix_fmt="yuv444p10le"),
    "no_overwrite_no_inherit": dict(video_codec="libx264", overwrite=False, inherit_color_metadata=False,
                                    audio_codec="copy"),
}

LUTS = {"none": None, "plain": "look.cube", "quoted": "/l u't/a.cube", "win": "C:\\luts\\it's.cube"}


def run_case(pname, iname, lname):
    params = ProcessingParams(**PARAM_SETS[pname])
    info = VideoInfo(**INFOS[iname]) if INFOS[iname] is not None else None
    lut = Path(LUTS[lname]) if LUTS[lname] else None
    notes = []
    case = {"params": PARAM_SETS[pname], "info": INFOS[iname], "lut": LUTS[lname]}
    try:
        case["argv"] = build_command(Path("in.mov"), Path("out.mp4"), params, lut_path=lut, source_info=info,
                                     notes=notes)
        case["notes"] = notes
    except ValueError as exc:
        case["error"] = str(exc)
    return case


def main():
    cases = {}
    for pname in PARAM_SETS:
        for iname in INFOS:
            for lname in ("plain", "none"):
                cases[f"{pname}|{iname}|{lname}"] = run_case(pname, iname, lname)
    for lname in ("quoted", "win"):
        cases[f"x265|tv10|{lname}"] = run_case("x265", "tv10", lname)
    pipelines = {}
    for mode, inter in (("fast", None), ("pro", "/m/in_master.mov"), ("pro", None)):
        params = ProcessingParams(video_codec="libx264", processing_mode=mode, crf="20", audio_bitrate="128k")
        task = Task("t1", Path("in.mov"), Path("out.mp4"), Path("look.cube"), None, params,
                    intermediate_path=Path(inter) if inter else None)
        key = f"{mode}|{'inter' if inter else 'nointer'}"
        try:
            stages = build_pipeline(task)
            pipelines[key] = {"stages": [
                {"name": s.name, "source": str(s.source_path), "output": str(s.output_path),
                 "lut": str(s.lut_path) if s.lut_path else None, "cleanup": s.cleanup_on_success,
                 "notes": list(s.notes), "probe": s.probe_source, "params": s.params.to_dict()} for s in stages]}
        except ValueError as exc:
            pipelines[key] = {"error": str(exc)}
    out = {"generator": "tests/golden/make_argv_fixtures.py (imports /root/reference/src/lut_renderer/ffmpeg.py)",
           "escape": {p: _escape_filter_path(Path(p)) for p in ("/l u't/a.cube", "C:\\luts\\it's.cube", "plain.cube")},
           "params_defaults": ProcessingParams().to_dict(),
           "videoinfo_fields": [f.name for f in dataclasses.fields(VideoInfo)],
           "cases": cases, "pipelines": pipelines}
    path = Path(__file__).with_name("argv_cases.json")
    path.write_text(json.dumps(out, ensure_ascii=False, indent=1, sort_keys=True) + "\n")
    print(f"wrote {path} with {len(cases)} argv cases and {len(pipelines)} pipelines")


if __name__ == "__main__":
    main()
