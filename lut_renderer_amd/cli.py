"""Engine CLI with the process contract `TaskRunner._run_stage` expects from ffmpeg
(/root/reference/src/lut_renderer/task_manager.py:134-190): started with Popen(stdout=PIPE,
stderr=STDOUT, text=True), it prints one `Duration: HH:MM:SS.xx` line, then `time=HH:MM:SS.xx`
progress lines (regexes at task_manager.py:14-15), exits 0 on success and non-zero with a message
on error, and stops on SIGTERM (task_manager.py:38-44).

It works on rawvideo files (planar frames back to back), i.e. the stage between a decoder and an
encoder; the options are the reference's own LUT vocabulary (models.py:45-56):

    python -m lut_renderer_amd.cli -i in.yuv -o out.yuv --size 3840x2160 --pix-fmt yuv420p10le \
        --cube look.cube --interp tetrahedral --colorspace bt2020nc --color-range tv

SURVEY.md 8f rank 1.  Frames stream through `stream.HostPipeline` (pinned ring, overlapped copies).
"""
from __future__ import annotations

import argparse
import os
import signal
import sys
import time


def _hms(seconds: float) -> str:
    seconds = max(0.0, seconds)
    h, rem = divmod(seconds, 3600)
    m, s = divmod(rem, 60)
    return f"{int(h):02d}:{int(m):02d}:{s:05.2f}"


def _rate(text: str) -> float:
    """'25', '29.97' or ffprobe's rational form '30000/1001'."""
    num, _, den = str(text).partition("/")
    value = float(num) / (float(den) if den else 1.0)
    if not value > 0:
        raise argparse.ArgumentTypeError(f"bad frame rate '{text}'")
    return value


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="lut_renderer_amd.cli", description=__doc__.split("\n\n")[0])
    ap.add_argument("-i", "--input", required=True)
    ap.add_argument("-o", "--output", required=True)
    ap.add_argument("--size", required=True, help="WxH")
    ap.add_argument("--pix-fmt", required=True)
    ap.add_argument("--out-pix-fmt", default=None)
    ap.add_argument("--cube", required=True)
    ap.add_argument("--interp", default="tetrahedral")
    ap.add_argument("--input-matrix", default="auto")
    ap.add_argument("--output-tags", default="bt709")
    ap.add_argument("--zscale-dither", default="none", help="none | error_diffusion (models.py:46)")
    ap.add_argument("--colorspace", default=None)
    ap.add_argument("--color-range", default=None)
    ap.add_argument("--fps", type=_rate, default=25.0, help="frame rate, a number or a rational like 30000/1001")
    ap.add_argument("--precision", default="strict", choices=["strict", "fast"],
                    help="engine setting (not one of the reference's options): strict = bit-exact fp32 restatement of FFmpeg's "
                         "scalar C (default); fast = tolerance-bounded kernels with an fp16 lattice, <= 1 code from strict")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("-y", action="store_true", help="overwrite the output (ffmpeg's -y)")
    ap.add_argument("--duration", type=float, default=None,
                    help="seconds of video, for the Duration: line when the input is a pipe (-i -) and cannot be measured")
    return ap


def plan_from_args(args):
    """The LutPlan and engine call the options select -- the same resolution `build_command` performs (plan.py)."""
    from .api import engine_call_for
    from .params import ProcessingParams, VideoInfo, infer_bit_depth
    from .plan import resolve_lut_plan
    w, h = (int(v) for v in args.size.lower().split("x"))
    params = ProcessingParams(lut_interp=args.interp, lut_input_matrix=args.input_matrix,
                              lut_output_tags=args.output_tags, zscale_dither=args.zscale_dither)
    info = VideoInfo(width=w, height=h, pix_fmt=args.pix_fmt, bit_depth=infer_bit_depth(args.pix_fmt),
                     colorspace=args.colorspace, color_range=args.color_range)
    plan = resolve_lut_plan(params, args.cube, info)
    kw = engine_call_for(plan, args.pix_fmt, args.out_pix_fmt)
    if args.zscale_dither == "error_diffusion":
        kw["dither"] = "error_diffusion"
    return plan, kw, w, h


def main(argv=None) -> int:
    args = build_parser().parse_args(argv)

    stop = {"flag": False}
    signal.signal(signal.SIGTERM, lambda *_: stop.__setitem__("flag", True))
    try:
        piped_in, piped_out = args.input == "-", args.output == "-"
        # with frames on stdout the report goes to stderr (the caller merges the two, task_manager.py:145-151)
        report = sys.stderr if piped_out else sys.stdout

        def say(text):
            print(text, file=report, flush=True)

        if not piped_out and os.path.exists(args.output) and not args.y:
            raise FileExistsError(f"{args.output} exists (pass -y to overwrite)")
        plan, kw, w, h = plan_from_args(args)
        from .cube import read_lut
        from .engine import LutEngine
        from .stream import HostPipeline

        eng = LutEngine(args.device)
        eng.set_precision(args.precision)
        eng.set_lut(read_lut(args.cube))
        pix_fmt, out_fmt = kw.pop("pix_fmt"), kw.pop("out_pix_fmt")
        pipe = HostPipeline(eng, pix_fmt, w, h, batch=args.batch, out_pix_fmt=out_fmt, **kw)
        fb = pipe.fin.frame_bytes
        if piped_in:
            total = None if args.duration is None else max(1, int(round(args.duration * args.fps)))
        else:
            total = os.path.getsize(args.input) // fb
            if total == 0:
                raise ValueError(f"{args.input}: no complete {w}x{h} {args.pix_fmt} frame ({fb} bytes each)")
        say(f"Input #0, rawvideo, from '{'pipe:0' if piped_in else args.input}':")
        if total is not None:
            say(f"  Duration: {_hms(total / args.fps)}, {total} frames, {w}x{h} {args.pix_fmt}")
        else:
            say(f"  Duration: N/A, {w}x{h} {args.pix_fmt}")
        for note in plan.notes:
            say(f"  {note}")
        t0 = time.time()
        state = {"done": 0}
        fi = sys.stdin.buffer if piped_in else open(args.input, "rb")
        fo = sys.stdout.buffer if piped_out else open(args.output, "wb")
        try:
            def fill(buf, max_frames):
                view, got = memoryview(buf)[: max_frames * fb], 0
                while got < len(view):                       # a pipe returns short reads: collect the whole batch (or EOF)
                    n = fi.readinto(view[got:])
                    if not n:
                        break
                    got += n
                return got // fb                             # a trailing partial frame is dropped

            def drain(buf, n):
                fo.write(memoryview(buf))
                state["done"] += n
                el = max(time.time() - t0, 1e-9)
                say(f"frame={state['done']:6d} fps={state['done'] / el:7.1f} time={_hms(state['done'] / args.fps)}")

            pipe.run(fill, drain, total_frames=None if piped_in else total, stop=lambda: stop["flag"])
            fo.flush()
        finally:
            if not piped_in:
                fi.close()
            if not piped_out:
                fo.close()
        eng.close()
        if stop["flag"]:
            say("Exiting normally, received signal 15.")
            return 255
        say(f"video: {state['done']} frames written to '{'pipe:1' if piped_out else args.output}'")
        return 0
    except Exception as exc:  # the caller only sees text + exit code (task_manager.py:105-112)
        print(f"Error: {exc}", file=sys.stderr if getattr(args, "output", "") == "-" else sys.stdout, flush=True)
        return 1


if __name__ == "__main__":
    sys.exit(main())
