"""decode -> raw pipe -> engine -> raw pipe -> encode: the LUT stage of a task run on the GPU engine with ffmpeg on both sides.

The reference runs ONE ffmpeg process per stage whose filtergraph does decode, `lut3d` and encode together
(`/root/reference/src/lut_renderer/ffmpeg.py:179-414`, spawned at `task_manager.py:134-190`).  With the per-pixel work moved to
`liblutr.so`, the same stage is three processes joined by OS pipes:

    ffmpeg -i SRC -f rawvideo -pix_fmt <src fmt> pipe:1          (decoder: the reference's input options, no filters)
      | python -m lut_renderer_amd.cli -i - -o - ...               (engine: `command.engine_command`, the same LutPlan)
      | ffmpeg -f rawvideo -pix_fmt <out fmt> -s WxH -r NUM/DEN -i pipe:0 -i SRC -map 0:v:0 -map 1:a? -map 1:s?
               -map_metadata 1 -map_chapters 1 <the reference's codec / rate / tag options> OUT

The encoder reads the source a second time for everything that is not video: the reference's single ffmpeg process carries the
source's audio (its `-c:a copy` / aac options), subtitles, chapters and container metadata into the output, and a raw pipe has
none of them.  The frame rate travels as the rational ffprobe reported (30000/1001, not 29.97).

`engine_stage_commands` derives all three argv lists from the arguments `build_command` takes; the encoder's options are what
`build_command` itself emits once the `-vf` chain is taken out (the engine has already applied it, including `format=`).
`run_stage` starts them, relays the engine's `Duration:` / `time=` lines and the three exit codes the way `TaskRunner._run_stage`
expects from a single child (non-zero if any stage failed; SIGTERM stops all three), so
`python -m lut_renderer_amd.pipe ...` can stand where `ffmpeg` stands in `CommandStage`.

There is no ffmpeg binary in this image: the tests drive the wrapper with `cat` on both sides.
"""
from __future__ import annotations

import signal
import subprocess
import sys
import threading
from dataclasses import dataclass
from pathlib import Path
from typing import List, Optional

from .command import build_command, engine_command, fps_rational
from .params import ProcessingParams, VideoInfo
from .plan import resolve_pix_fmt


@dataclass
class StageCommands:
    decoder: List[str]
    engine: List[str]
    encoder: List[str]
    notes: List[str]


def engine_stage_commands(source: Path, output: Path, params: ProcessingParams, lut_path: Path, source_info: VideoInfo,
                          ffmpeg_bin: str = "ffmpeg", python_bin: Optional[str] = None, device: int = 0,
                          precision: str = "strict") -> StageCommands:
    """The three argv lists of one LUT stage.  Raises what `build_command` / `engine_command` raise (copy guard, missing
    geometry)."""
    notes: List[str] = []
    engine = engine_command(Path("-"), Path("-"), params, lut_path, source_info, python_bin=python_bin, device=device, notes=notes,
                            precision=precision)
    if source_info.duration:
        engine += ["--duration", f"{float(source_info.duration):.3f}"]
    decoder = [ffmpeg_bin, "-hide_banner", "-nostdin", "-i", str(source), "-map", "0:v:0", "-f", "rawvideo",
               "-pix_fmt", str(source_info.pix_fmt), "pipe:1"]
    # the encoder side: build_command's own argv for this task WITHOUT a LUT (no -vf chain), reading raw frames from the pipe
    out_fmt = resolve_pix_fmt(params, source_info, []) if params.video_codec else ""
    if not out_fmt:
        out_fmt = str(source_info.pix_fmt)
        if engine.count("--out-pix-fmt"):
            out_fmt = engine[engine.index("--out-pix-fmt") + 1]
    raw_in = ["-f", "rawvideo", "-pix_fmt", out_fmt, "-s", f"{source_info.width}x{source_info.height}"]
    if source_info.fps:
        raw_in += ["-r", fps_rational(source_info.fps)]
    enc_notes: List[str] = []
    tail = build_command(Path("pipe:0"), output, params, lut_path=None, ffmpeg_bin=ffmpeg_bin, source_info=source_info,
                         notes=enc_notes)
    i = tail.index("-i")
    # input 0 = the engine's frames, input 1 = the source again for audio / subtitles / chapters / metadata (`?`: optional
    # streams; video is taken from the pipe only, so the source's picture is never decoded a second time)
    side = ["-i", str(source), "-map", "0:v:0", "-map", "1:a?", "-map", "1:s?", "-map_metadata", "1", "-map_chapters", "1"]
    encoder = tail[:i] + raw_in + tail[i:i + 2] + side + tail[i + 2:]
    # colour tags are decided by the LUT policy (ffmpeg.py:348-383), which build_command only applies with a lut_path:
    # take them from the full command
    full = build_command(source, output, params, lut_path=lut_path, ffmpeg_bin=ffmpeg_bin, source_info=source_info, notes=[])
    for flag in ("-color_primaries", "-color_trc", "-colorspace", "-color_range"):
        if flag in encoder:
            j = encoder.index(flag)
            del encoder[j:j + 2]
    for flag in ("-color_primaries", "-color_trc", "-colorspace", "-color_range"):
        if flag in full:
            encoder[-1:-1] = [flag, full[full.index(flag) + 1]]
    return StageCommands(decoder, engine, encoder, notes)


def run_stage(cmds: StageCommands, out=sys.stdout) -> int:
    """Run decoder | engine | encoder.  The engine's report (its stderr, since its stdout carries frames) is relayed line by
    line to `out`, which is what `TaskRunner._run_stage` parses; returns 0 only if all three exit 0."""
    dec = subprocess.Popen(cmds.decoder, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    eng = subprocess.Popen(cmds.engine, stdin=dec.stdout, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    enc = subprocess.Popen(cmds.encoder, stdin=eng.stdout, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dec.stdout.close()          # the readers own the pipe ends now: a writer gets SIGPIPE when its reader dies
    eng.stdout.close()
    procs = [dec, eng, enc]

    def stop(*_):
        for p in procs:
            if p.poll() is None:
                p.terminate()

    old = signal.signal(signal.SIGTERM, stop) if threading.current_thread() is threading.main_thread() else None
    try:
        for raw in eng.stderr:
            out.write(raw.decode("utf-8", "replace"))
            out.flush()
        codes = [p.wait() for p in procs]
    finally:
        if old is not None:
            signal.signal(signal.SIGTERM, old)
    for name, code in zip(("decoder", "engine", "encoder"), codes):
        if code:
            out.write(f"Error: {name} exited with {code}\n")
            out.flush()
            return code if code > 0 else 255
    return 0


def main(argv=None) -> int:
    import argparse
    import json
    ap = argparse.ArgumentParser(prog="lut_renderer_amd.pipe", description=__doc__.split("\n\n")[0])
    ap.add_argument("-i", "--input", required=True)
    ap.add_argument("-o", "--output", required=True)
    ap.add_argument("--cube", required=True)
    ap.add_argument("--params", default="{}", help="ProcessingParams as JSON (models.py:58-122 dict form)")
    ap.add_argument("--info", required=True, help="VideoInfo fields as JSON: width, height, pix_fmt, fps, colorspace, ...")
    ap.add_argument("--ffmpeg", default="ffmpeg")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--precision", default="strict", choices=["strict", "fast"], help="engine setting, see lut_renderer_amd.cli")
    a = ap.parse_args(argv)
    try:
        params = ProcessingParams.from_dict(json.loads(a.params))
        info = VideoInfo(**json.loads(a.info))
        cmds = engine_stage_commands(Path(a.input), Path(a.output), params, Path(a.cube), info, ffmpeg_bin=a.ffmpeg, device=a.device,
                                     precision=a.precision)
    except Exception as exc:
        print(f"Error: {exc}", flush=True)
        return 1
    return run_stage(cmds)


if __name__ == "__main__":
    sys.exit(main())
