"""Drop-in host layer vs argv/notes/errors captured from the reference's own builder.

tests/golden/argv_cases.json was produced by tests/golden/make_argv_fixtures.py, which imports
/root/reference/src/lut_renderer/ffmpeg.py (build_command :179-414, build_pipeline :436-487)
in the build container.  These tests read only the JSON, so they also run where the
reference does not exist.  Covers SURVEY.md 8a rows a1-a13 and Appendix D cases A-P.
"""
import dataclasses
import json
from pathlib import Path

import pytest

from lut_renderer_amd.command import build_command, build_pipeline
from lut_renderer_amd.params import ProcessingParams, Task, VideoInfo
from lut_renderer_amd.plan import escape_filter_path, resolve_lut_plan

GOLD = json.loads((Path(__file__).parent / "golden" / "argv_cases.json").read_text())


@pytest.mark.parametrize("key", sorted(GOLD["cases"]))
def test_build_command_matches_reference(key):
    case = GOLD["cases"][key]
    params = ProcessingParams(**case["params"])
    info = VideoInfo(**case["info"]) if case["info"] is not None else None
    lut = Path(case["lut"]) if case["lut"] else None
    notes = []
    if "error" in case:
        with pytest.raises(ValueError) as exc:
            build_command(Path("in.mov"), Path("out.mp4"), params, lut_path=lut, source_info=info, notes=notes)
        assert str(exc.value) == case["error"]
        return
    argv = build_command(Path("in.mov"), Path("out.mp4"), params, lut_path=lut, source_info=info, notes=notes)
    assert argv == case["argv"]
    assert notes == case["notes"]        # the out-parameter list is mutated in place, same strings


def test_pipelines_match_reference():
    for key, want in GOLD["pipelines"].items():
        mode, inter = key.split("|")
        params = ProcessingParams(video_codec="libx264", processing_mode=mode, crf="20", audio_bitrate="128k")
        task = Task("t1", Path("in.mov"), Path("out.mp4"), Path("look.cube"), None, params,
                    intermediate_path=Path("/m/in_master.mov") if inter == "inter" else None)
        if "error" in want:
            with pytest.raises(ValueError) as exc:
                build_pipeline(task)
            assert str(exc.value) == want["error"]
            continue
        stages = build_pipeline(task)
        got = [{"name": s.name, "source": str(s.source_path), "output": str(s.output_path),
                "lut": str(s.lut_path) if s.lut_path else None, "cleanup": s.cleanup_on_success,
                "notes": list(s.notes), "probe": s.probe_source, "params": s.params.to_dict()} for s in stages]
        assert got == want["stages"]


def test_records_have_the_reference_shape():
    assert ProcessingParams().to_dict() == GOLD["params_defaults"]
    assert [f.name for f in dataclasses.fields(VideoInfo)] == GOLD["videoinfo_fields"]
    # dict round trip, including bool coercion (models.py:109-117)
    p = ProcessingParams.from_dict({"lut_interp": "trilinear", "faststart": 1, "unknown_key": 3})
    assert p.lut_interp == "trilinear" and p.faststart is True
    assert ProcessingParams.from_dict(p.to_dict()) == p


def test_filter_path_escaping():
    for raw, want in GOLD["escape"].items():
        assert escape_filter_path(Path(raw)) == want


def test_smoke_asserts_of_the_reference():
    """The three checks of /root/reference/src/lut_renderer/smoke.py:21-43."""
    with pytest.raises(ValueError):
        build_command(Path("input.mov"), Path("output.mp4"), ProcessingParams(video_codec="copy"),
                      lut_path=Path("look.cube"))
    ten = VideoInfo(bit_depth=10, pix_fmt="yuv420p10le")
    cmd = build_command(Path("i"), Path("o"), ProcessingParams(video_codec="libx265"), source_info=ten)
    assert "-pix_fmt yuv420p10le" in " ".join(cmd)
    cmd = build_command(Path("i"), Path("o"), ProcessingParams(video_codec="libx264", bit_depth_policy="force_8bit"),
                        lut_path=Path("look.cube"), source_info=ten)
    joined = " ".join(cmd)
    for frag in ("-color_primaries bt709", "-color_trc bt709", "-colorspace bt709", "-color_range tv"):
        assert frag in joined


def test_plan_to_engine_call():
    """The same plan drives the engine: prologue -> 8-bit LUT, matrix fallback BT.601, tv output."""
    from lut_renderer_amd.api import engine_call_for
    pc10 = VideoInfo(bit_depth=10, pix_fmt="yuv422p10le", color_range="pc", colorspace="bt2020nc")
    plan = resolve_lut_plan(ProcessingParams(lut_input_matrix="none"), "look.cube", pc10)
    kw = engine_call_for(plan, "yuv422p10le", out_pix_fmt="yuv422p10le")
    assert kw["lut_depth"] == 8 and kw["range_src"] == "pc" and kw["range_in"] == "tv"
    assert kw["matrix_in"] == "smpte170m" and kw["range_out"] == "tv" and kw["out_pix_fmt"] == "yuv422p10le"
    tv10 = VideoInfo(bit_depth=10, pix_fmt="yuv420p10le", color_range="tv", colorspace="bt2020nc")
    kw = engine_call_for(resolve_lut_plan(ProcessingParams(), "look.cube", tv10), "yuv420p10le")
    assert kw["lut_depth"] == 10 and kw["matrix_in"] == "bt2020nc" and kw["out_pix_fmt"] == "yuv420p10le"
    assert kw["interp"] == "tetrahedral"
    with pytest.raises(ValueError):
        engine_call_for(resolve_lut_plan(ProcessingParams(lut_interp="cubic"), "l.cube", tv10), "yuv420p10le")
    j8 = VideoInfo(bit_depth=8, pix_fmt="yuvj420p", color_range="pc", colorspace="bt709")
    kw = engine_call_for(resolve_lut_plan(ProcessingParams(lut_output_tags="inherit"), "l.cube", j8), "yuvj420p")
    assert kw["pix_fmt"] == "yuv420p" and kw["range_in"] == "pc" and kw["out_pix_fmt"] == "yuv420p"
