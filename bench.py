#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the 3D-LUT apply hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment the script starts its own N ranks (one per GPU,
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` as a child
process, before this process has touched the GPU) and relays rank 0's JSON line; launched by
torch.distributed.run itself it is one of the ranks.

A "step" is one pass of the fused kernel over a batch of synthetic frames that is already resident in
HBM.  Default workload = BASELINE.json configs[1]: 3840x2160 yuv420p10le, 33^3 log->Rec.709 .cube,
tetrahedral, 256 frames per GPU and step (12.7 GB of the 288 GB HBM).

Multi-GPU: every frame is split into N row blocks with FFmpeg's slice rule; rank g owns
block g and launches on the SHARED full-height planes with row0 = its first row.  `value` is weak
scaling (N x FRAMES frames, so per-GPU work is fixed); `strong` in the same line is the row-shard
speed-up north_star words: the same FRAMES frames, rows split N ways.  The only collective is the RCCL
broadcast of the lattice at LUT load, reported in `collective`.

Rank 0 prints ONE JSON line.  `value`, `ms_per_step`, `roofline` and `dtype` describe the STRICT kernels
(`config.precision`): fp32 lattice, fp32 blend in FFmpeg's scalar-C order, bit-identical to the oracle.  The
tolerance-bounded FAST kernels (fp16 lattice, <= 1 code from strict) are timed on the same batch and reported
in `other_precision` with their own `kernel_ms` and dtype string -- never as `value`.
`roofline.achieved` = algorithmic bytes per launch (6 B/px for yuv420p10le in+out, + the 431,244 B lattice)
/ the kernel's mean launch duration measured with HIP events on the launch stream.  `cpu_baseline` times the
CPU oracle (kind "port": a restatement of FFmpeg lut3d, not FFmpeg) on the host cores.  `extra_Mpx_s` carries
both precisions on other frame statistics (three times the chroma, sensor noise up to sigma = 64 codes,
i.i.d. uniform) with the LDS-window hit / miss / gather counts.  `host_pipeline` is BASELINE config 5 (256 UHD
full-range frames queued in pinned host memory, pc->tv prologue fused, overlapped hipMemcpyAsync) with the
PCIe rates measured in the same run beside it; `host_us_per_apply` is the launcher's host cost per call.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

SIZES = {"1080p": (1920, 1080), "uhd": (3840, 2160), "8k": (7680, 4320)}
HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
EXTRA_DISTS = ("vivid", "noise8", "noise16", "noise64", "uniform")
# what the kernels of a precision compute in: `dtype` of the line is the arithmetic type, not a precision claim
DTYPE = {"strict": "f32", "fast": "f32 coordinates and accumulation, f16 lattice"}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames", type=int, default=256,
                    help="frames per GPU per step (SURVEY 8d: >= 64; 256 = the batch of BASELINE config 5: 6.4 GB in + 6.4 GB out)")
    ap.add_argument("--size", default="uhd", choices=sorted(SIZES))
    ap.add_argument("--fmt", default="yuv420p10le")
    ap.add_argument("--out-fmt", default=None, help="output pixel format (default: same as --fmt); e.g. yuv420p for the "
                                                    "reference's libx264 default on a 10-bit source (ffmpeg.py:287-302)")
    ap.add_argument("--range-src", default="tv", choices=["tv", "pc"],
                    help="pc: full-range source -> the reference's prologue scale=in_range=pc:out_range=tv,format=<8-bit> "
                         "(ffmpeg.py:212-233, BASELINE config 5) runs fused ahead of the LUT")
    ap.add_argument("--interp", default="tetrahedral")
    ap.add_argument("--precision", default="strict", choices=["strict", "fast"],
                    help="strict (default, named in config.precision): fp32 lattice and blend, the bit-exact restatement of "
                         "FFmpeg's scalar C -- the reference's precision; fast: the tolerance-bounded kernels (fp16 lattice, "
                         "<= 1 code from strict at 8 and 10 bit, tests/test_fast_variant.py).  The other one is timed too and "
                         "reported in `other_precision`.")
    ap.add_argument("--lut", type=int, default=33, help="lattice size N of the generated log709 LUT")
    ap.add_argument("--dist", default="natural")
    ap.add_argument("--variant", default="auto")
    ap.add_argument("--unique", type=int, default=2, help="distinct synthetic frames tiled into the batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stats", action="store_true", help="skip the extra LDS-window statistics pass (profiling runs)")
    ap.add_argument("--no-extra", action="store_true", help="skip the other-content lines (extra_Mpx_s)")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling pass (N > 1)")
    ap.add_argument("--no-other", action="store_true", help="skip timing the other precision (profiling runs: one kernel only)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0,
                    help="wall budget of the CPU baseline sample (all host cores: ~10 s wall on 256 threads)")
    ap.add_argument("--extra-modes", action="store_true", help="extra_Mpx_s also for trilinear")
    ap.add_argument("--prewarm-ms", type=float, default=60.0,
                    help="untimed kernel launches before the W warm-up steps until this much time has passed: the GPU "
                         "needs ~20 ms of load to leave its idle clock (DESIGN.md 5), whatever W the caller picks")
    ap.add_argument("--dither", default="none", choices=["none", "error_diffusion"],
                    help="also dither the final quantisation (reference option zscale_dither; YUV formats, informational)")
    ap.add_argument("--pipeline", default="auto", choices=["auto", "hbm", "host"],
                    help="auto (default): after the HBM-resident metric also time BASELINE config 5 -- 256 full-range frames "
                         "in pinned host memory, pc->tv prologue + LUT, overlapped hipMemcpyAsync -- and the PCIe copy rates of "
                         "this box beside it; reported as `host_pipeline`, never as `value`.  host: the same leg with the "
                         "range / format options of this run instead of config 5's.  hbm: skip it")
    ap.add_argument("--host-frames", type=int, default=256)
    ap.add_argument("--no-host-cost", action="store_true", help="skip the host-microseconds-per-apply measurement")
    ap.add_argument("--lean", action="store_true",
                    help="the timed kernel(s) only: --no-cpu-baseline --no-extra --no-strong --pipeline hbm --no-host-cost (tools/)")
    args = ap.parse_args(argv)
    if args.lean:
        args.no_cpu_baseline = args.no_extra = args.no_strong = args.no_host_cost = True
        args.pipeline = "hbm"
    return args


# ---------------------------------------------------------------- self-launch (N > 1 without a launcher)
def spawn_ranks(n: int) -> int:
    """Start N fresh ranks as a child process tree and relay their output.  This process has not initialised the
    GPU (no torch.cuda call so far) and never replaces itself: it waits for the child and returns its exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    log(f"[bench] starting {n} ranks: {' '.join(cmd)}")
    return subprocess.call(cmd, env=env)


def dist_setup(gpus: int):
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal knobs (tests/test_gpu_parity.py): run the N-rank control flow on a box with fewer GPUs by
    # putting every rank on one device and moving the collectives to gloo.  Never set by the driver.
    backend = os.environ.get("LUTR_DIST_BACKEND", "nccl")
    if "LUTR_FORCE_DEVICE" in os.environ:
        local = int(os.environ["LUTR_FORCE_DEVICE"])
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
        assert dist.get_world_size() == gpus, f"--gpus {gpus} but WORLD_SIZE {world}"
    return rank, local, world, backend


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


class Job:
    """What one timed launch does: format pair, mode, prologue, row range."""

    def __init__(self, args, pf, pf_out):
        self.args, self.pf, self.pf_out = args, pf, pf_out
        self.kw = {}
        if pf.family == "yuv":
            self.kw = dict(pix_fmt=args.fmt, out_pix_fmt=args.out_fmt or args.fmt)
            if args.range_src == "pc":
                # ffmpeg.py:212-233: scale=in_range=pc:out_range=tv , format=<8-bit intermediate>: the LUT runs at 8 bit
                self.kw.update(range_src="pc", range_in="tv", lut_depth=8)
            if args.dither != "none":
                self.kw["dither"] = args.dither

    def apply(self, eng, src, dst, interp, row0=0, rows=None):
        pf = self.pf
        if pf.family == "packed":
            return eng.apply_packed(src[0], dst[0], pix_fmt=self.args.fmt, interp=interp, row0=row0, rows=rows)
        if pf.family == "gbr":
            return eng.apply_rgb(src, dst, depth=pf.depth, interp=interp, row0=row0, rows=rows)
        if self.args.dither != "none":
            return eng.apply_yuv(src, dst, interp=interp, **self.kw)
        return eng.apply_yuv(src, dst, interp=interp, row0=row0, rows=rows, **self.kw)


def make_frames(pf, w, h, dist_name, k, full_range):
    from lut_renderer_amd import frames
    if pf.family in ("gbr", "packed"):
        return frames.make_rgb(dist_name, w, h, pf.depth, k=k)
    return frames.make_yuv(dist_name, w, h, pf.depth, pf.csx, pf.csy, k=k, full_range=full_range)


def build_batch(eng, pf, w, h, nframes, dist_name, unique, full_range=False):
    """`unique` synthetic full-height frames tiled to `nframes` frames on the device."""
    import numpy as np
    import torch
    reps = (nframes + unique - 1) // unique
    if pf.family == "packed":                            # interleave the planar RGB generator's frames
        imgs = []
        for k in range(unique):
            g, b, r = make_frames(pf, w, h, dist_name, k, full_range)
            img = np.full((h, w, pf.nc), (1 << pf.depth) - 1, dtype=g.dtype)
            img[..., pf.rgb[0]], img[..., pf.rgb[1]], img[..., pf.rgb[2]] = r, g, b
            imgs.append(torch.from_numpy(img.view(np.int16) if img.dtype == np.uint16 else img))
        u = torch.stack(imgs).to(eng.device)
        return [u.repeat(reps, 1, 1, 1)[:nframes].contiguous()]
    planes = [[], [], []]
    for k in range(unique):
        f = make_frames(pf, w, h, dist_name, k, full_range)
        for i in range(3):
            a = np.ascontiguousarray(f[i])
            planes[i].append(torch.from_numpy(a.view(np.int16) if a.dtype == np.uint16 else a))
    out = []
    for i in range(3):
        u = torch.stack(planes[i]).to(eng.device)
        out.append(u.repeat(reps, 1, 1)[:nframes].contiguous())
    return out


def alloc_out(eng, job, src, w, h):
    import torch
    pf, pfo = job.pf, job.pf_out
    if pf.family != "yuv" or pfo is pf:
        return [torch.empty_like(t) for t in src]
    dt = torch.uint8 if pfo.depth <= 8 else torch.int16
    nf = src[0].shape[0]
    return [torch.empty((nf,) + pfo.plane_shape(i, w, h), dtype=dt, device=eng.device) for i in range(3)]


PREWARM_MS = 0.0


def time_steps(eng, job, src, dst, interp, steps, warmup, world, row0=0, rows=None, nframes=None):
    """Returns (wall seconds for `steps` launches, mean kernel seconds from HIP events on the launch stream)."""
    import torch
    if nframes is not None:
        src, dst = [t[:nframes] for t in src], [t[:nframes] for t in dst]
    t_end = time.perf_counter() + PREWARM_MS / 1e3
    while time.perf_counter() < t_end:                 # clock ramp; not part of W, not timed
        job.apply(eng, src, dst, interp, row0, rows)
        torch.cuda.synchronize()
    for _ in range(warmup):
        job.apply(eng, src, dst, interp, row0, rows)
    torch.cuda.synchronize()
    barrier(world)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        job.apply(eng, src, dst, interp, row0, rows)
    ev1.record()
    torch.cuda.synchronize()
    barrier(world)
    wall = time.perf_counter() - t0
    return wall, ev0.elapsed_time(ev1) / 1e3 / steps


def reduce_max_sum(eng, world, maxes, sums):
    import torch
    if world == 1:
        return list(maxes), list(sums)
    import torch.distributed as dist
    dev = eng.device if dist.get_backend() == "nccl" else "cpu"
    tm = torch.tensor(maxes, dtype=torch.float64, device=dev)
    ts = torch.tensor(sums, dtype=torch.float64, device=dev)
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    dist.all_reduce(ts, op=dist.ReduceOp.SUM)
    return tm.tolist(), ts.tolist()


def cpu_baseline(lut, job, w, h, interp, dist_name, budget_s):
    """CPU oracle (port of FFmpeg lut3d + this repo's YUV contract) on all host cores, row-sliced
    like FFmpeg's slice threads; bounded sample of the same workload."""
    from oracle import binding as orc
    pf, pfo, args = job.pf, job.pf_out, job.args
    cores = os.cpu_count() or 1
    f = make_frames(pf, w, h, dist_name, 0, args.range_src == "pc")
    if pf.family == "gbr":
        def run():
            orc.apply_rgb(lut.table, lut.scale, pf.depth, interp, f, nthreads=cores)
    else:
        ld = 8 if args.range_src == "pc" else pf.depth
        k = orc.yuv_constants("bt709", "tv", "bt709", "tv", pf.depth, ld, pfo.depth, 1 << (pf.csx + pf.csy),
                              prologue=args.range_src == "pc")

        def run():
            orc.apply_yuv(lut.table, lut.scale, interp, k, pf.depth, ld, pfo.depth, pf.csx, pf.csy, f, nthreads=cores)
    run()
    n, t0 = 0, time.perf_counter()
    while True:
        run()
        n += 1
        el = time.perf_counter() - t0
        if n >= 2 and el >= budget_s:
            break
    return {"value": round(n * w * h / el / 1e6, 1), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": f"{n} x {w}x{h} {pf.name} frames ({dist_name}), {interp}, {cores} row-slice threads, "
                      f"{el:.1f} s wall; CPU restatement of FFmpeg lut3d (no ffmpeg binary on this image)"}


def pcie_rates(device, nbytes=256 << 20, reps=3):
    """hipMemcpyAsync rates of THIS box, pinned host memory, GB/s: H2D alone, D2H alone, and both directions at once on two
    streams (what the ring of `stream.HostPipeline` does).  The host pipeline's rate is judged against these, not a spec."""
    import torch
    hin, hout = torch.empty(nbytes, dtype=torch.uint8).pin_memory(), torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    din, dout = torch.empty(nbytes, dtype=torch.uint8, device=device), torch.empty(nbytes, dtype=torch.uint8, device=device)
    s1, s2 = torch.cuda.Stream(device), torch.cuda.Stream(device)

    def run(h2d, d2h):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            if h2d:
                with torch.cuda.stream(s1):
                    din.copy_(hin, non_blocking=True)
            if d2h:
                with torch.cuda.stream(s2):
                    hout.copy_(dout, non_blocking=True)
        s1.synchronize(); s2.synchronize()
        return reps * nbytes / (time.perf_counter() - t0) / 1e9

    run(True, True)
    return {"h2d_alone": round(run(True, False), 1), "d2h_alone": round(run(False, True), 1),
            "each_way_concurrent": round(run(True, True), 1), "copy_bytes": nbytes}


def host_pipeline_leg(eng, args, job, pf, w, h, rank, world):
    """BASELINE config 5: frames queued in pinned host memory, round-robin over the GPUs (whole frames per rank, every rank
    drives its own 3-slot ring of 8-frame batches, H2D / kernel / D2H on separate HIP streams); total = all ranks' frames /
    the slowest rank's time.  `auto` runs config 5 itself (full-range source, pc->tv prologue fused ahead of the LUT) whatever
    the headline's range is; `host` keeps this run's options."""
    import numpy as np
    import torch
    from lut_renderer_amd.stream import HostPipeline
    if args.pipeline == "auto":
        kw, full_range = dict(range_src="pc", range_in="tv", lut_depth=8), True
    else:
        kw = {k: v for k, v in job.kw.items() if k not in ("pix_fmt", "out_pix_fmt")}
        full_range = args.range_src == "pc"
    pipe = HostPipeline(eng, args.fmt, w, h, batch=8, slots=3, out_pix_fmt=args.out_fmt, interp=args.interp, **kw)
    full = make_frames(pf, w, h, args.dist, 0, full_range)
    one = b"".join(np.ascontiguousarray(p).tobytes() for p in full)
    for sl in range(pipe.slots):                 # inputs pre-filled: the producer is not what is measured
        pipe.host_in(sl)[:] = np.frombuffer(one * pipe.batch, dtype=np.uint8)
    mine = args.host_frames // world + (1 if rank < args.host_frames % world else 0)
    pipe.run(lambda b, m: m, lambda b, k: None, total_frames=24)        # warm-up
    torch.cuda.synchronize()
    barrier(world)
    t0 = time.perf_counter()
    n_done = pipe.run(lambda b, m: m, lambda b, k: None, total_frames=mine)
    torch.cuda.synchronize()
    barrier(world)
    el = time.perf_counter() - t0
    kernel = eng.last_kernel
    rates = pcie_rates(eng.device) if rank == 0 else None
    (el,), (n_all,) = reduce_max_sum(eng, world, [el], [float(n_done)])
    n_all = int(n_all)
    gb = n_all * (pipe.fin.frame_bytes + pipe.fout.frame_bytes) / 1e9
    each_way = gb / 2 / el
    out = {"frames": n_all, "fps": round(n_all / el, 1), "Mpixels_s": round(n_all * w * h / el / 1e6, 1),
           "pcie_GBps_each_way": round(each_way, 1), "gpus": world, "kernel": kernel,
           "prologue": "scale=in_range=pc:out_range=tv,format=<8-bit> fused" if full_range else "none",
           "note": "frames in pinned host memory, round-robin over the GPUs, per GPU a 3-slot ring of 8-frame batches with "
                   "H2D / kernel / D2H on separate streams; pcie_GBps_each_way is the sum over GPUs"}
    if rates:
        out["pcie_measured_GBps"] = rates
        out["frac_of_measured_pcie"] = round(each_way / world / rates["each_way_concurrent"], 3)
    return out


def host_us_per_apply(eng, job, args, n=1000):
    """Host cost of one apply: `n` launches of ONE 1080p frame (same format / chain as the headline) issued back to back
    without waiting.  `python` = wall per `LutEngine.apply_yuv` call (argument checks, ctypes, the C launcher);
    `c_abi` = wall per bare `lutr_apply_yuv` call with prebuilt descriptors (the launcher alone); `gpu` = the launch's
    duration on the device -- when it exceeds the host figure the queue, not the host, paces a stream of such calls."""
    import ctypes as C
    import torch
    from lut_renderer_amd import _native
    from lut_renderer_amd.engine import _planes_struct
    pf = job.pf
    w, h = 1920, 1080
    src = build_batch(eng, pf, w, h, 1, args.dist, 1, args.range_src == "pc")
    dst = alloc_out(eng, job, src, w, h)
    out = {}
    for variant in ("auto", "vec_lds"):
        eng.set_variant(variant)
        for _ in range(20):
            job.apply(eng, src, dst, args.interp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            job.apply(eng, src, dst, args.interp)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        # the same through the bare C-ABI: descriptors built once
        s, _ = _planes_struct(src, eng.device)
        d, _ = _planes_struct(dst, eng.device)
        from lut_renderer_amd.engine import parse_pix_fmt
        fin, fout = parse_pix_fmt(job.kw["pix_fmt"]), parse_pix_fmt(job.kw["out_pix_fmt"])
        prm = _native.YuvParams()
        prm.fmt_in, prm.fmt_out = fin.code, fout.code
        prm.lut_depth = job.kw.get("lut_depth", fin.depth)
        prm.matrix_in = prm.matrix_out = _native.MATRIX["bt709"]
        prm.range_src = _native.RANGE[job.kw.get("range_src", "tv")]
        prm.range_in = _native.RANGE[job.kw.get("range_in", job.kw.get("range_src", "tv"))]
        prm.range_out = _native.RANGE["tv"]
        lib, ctx, mode = eng._lib, eng._ctx, _native.INTERP[args.interp]
        eng._bind_stream()
        call = lib.lutr_apply_yuv
        ps, pd, pp = C.byref(s), C.byref(d), C.byref(prm)
        for _ in range(20):
            _native.check(call(ctx, pp, mode, w, h, 1, ps, pd, 0, h))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            call(ctx, pp, mode, w, h, 1, ps, pd, 0, h)
        t_c = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_c_all = time.perf_counter() - t0
        out[variant] = {"kernel": eng.last_kernel, "python_us": round(t_issue / n * 1e6, 2), "c_abi_us": round(t_c / n * 1e6, 2),
                        "gpu_us": round(min(t_all, t_c_all) / n * 1e6, 2)}
    eng.set_variant(args.variant)
    out["sample"] = f"{n} applies of one 1920x1080 {pf.name} frame, issued back to back on one stream"
    return out


def load_traffic(tag, name="traffic.json"):
    """HBM bytes per launch (or, with issue.json, the VALU issue figures) from the committed rocprofv3 --pmc summary, if one exists
    for this workload."""
    p = ROOT / "profiles" / name
    if p.exists():
        try:
            return json.loads(p.read_text()).get(tag)
        except Exception:
            return None
    return None


def bytes_per_px(pf):
    base = (float(pf.nc) if pf.family == "packed" else 3.0 if pf.family == "gbr"
            else 1.0 + 2.0 / (1 << (pf.csx + pf.csy)))
    return base * (1 if pf.depth <= 8 else 2)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))       # before anything here touches the GPU

    import numpy as np  # noqa: F401
    import torch
    global PREWARM_MS
    PREWARM_MS = args.prewarm_ms
    rank, local, world, backend = dist_setup(args.gpus)
    from lut_renderer_amd import cube
    from lut_renderer_amd._native import PACKED_FORMATS
    from lut_renderer_amd.engine import LutEngine, parse_pix_fmt
    from lut_renderer_amd.shard import my_rows

    w, h = SIZES[args.size]
    if args.fmt in PACKED_FORMATS:
        from types import SimpleNamespace
        bits, nc, *rgb = PACKED_FORMATS[args.fmt]
        pf = SimpleNamespace(family="packed", depth=bits, csx=0, csy=0, nc=nc, rgb=rgb, name=args.fmt)
        pf_out = pf
    else:
        pf = parse_pix_fmt(args.fmt)
        pf_out = parse_pix_fmt(args.out_fmt) if args.out_fmt and args.out_fmt != args.fmt else pf
    job = Job(args, pf, pf_out)
    eng = LutEngine(local)
    eng.set_variant(args.variant)
    if hasattr(eng, "set_precision"):
        eng.set_precision(args.precision)
    elif args.precision != "strict":
        raise SystemExit("this build has no fast variant")

    # LUT: generated log->Rec.709 lattice written as a real .cube, parsed by liblutr on rank 0,
    # broadcast to the other ranks (the path's only collective)
    lut = None
    if rank == 0:
        with tempfile.TemporaryDirectory() as d:
            lut = cube.read_cube(cube.write_cube(Path(d) / f"log709_{args.lut}.cube", cube.log709_lattice(args.lut),
                                                 title=f"log709 {args.lut}"))
    collective = None
    if world > 1:
        eng.set_lut_distributed(lut, src=0)                 # first use also builds the communicator
        torch.cuda.synchronize()
        barrier(world)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.set_lut_distributed(lut, src=0)                 # the steady-state cost of a LUT change
        e1.record()
        torch.cuda.synchronize()
        lat_bytes = eng.lattice_tensor().numel() * 4
        import torch.distributed as dist
        collective = {"backend": f"{dist.get_backend()}" + (" (RCCL over xGMI)" if dist.get_backend() == "nccl" else ""),
                      "world": dist.get_world_size(), "op": "broadcast(lattice), root 0, once per LUT load",
                      "bcast_bytes": int(lat_bytes + 16), "bcast_us": round(e0.elapsed_time(e1) * 1e3, 1),
                      "data_path_collectives": 0}
    else:
        eng.set_lut(lut)

    align = 1 << pf.csy
    r0, r1 = my_rows(h, rank, world, align=align)
    nframes = args.frames * world                     # weak scaling: row block g of world x FRAMES full-height frames
    full_range = args.range_src == "pc"
    t_setup = time.perf_counter()
    src = build_batch(eng, pf, w, h, nframes, args.dist, args.unique, full_range)
    dst = alloc_out(eng, job, src, w, h)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup
    if rank == 0:
        log(f"[setup] {nframes} frames of {w}x{h} {args.fmt} in + out on every rank: {t_setup:.2f} s")
    px_rank = (r1 - r0) * w * nframes

    wall, kern = time_steps(eng, job, src, dst, args.interp, args.steps, args.warmup, world, r0, r1 - r0)
    kernel_name = eng.last_kernel
    tile_stats = None
    if ("tile" in kernel_name or "tube" in kernel_name) and not args.no_stats:   # one extra, untimed pass with the window counters armed
        eng.tile_stats(True)
        job.apply(eng, src, dst, args.interp, r0, r1 - r0)
        tile_stats = eng.tile_stats(False)
        if rank == 0:
            log(f"[tile stats] {tile_stats}")
    (wall, kern), (px_total,) = reduce_max_sum(eng, world, [wall, kern], [float(px_rank)])

    # the other precision on the same batch (N = 1): the line always carries both numbers
    other = None
    if world == 1 and pf.family == "yuv" and args.dither == "none" and not args.no_other:
        oname = "strict" if args.precision == "fast" else "fast"
        eng.set_precision(oname)
        _, ok = time_steps(eng, job, src, dst, args.interp, max(5, args.steps // 4), 3, 1, r0, r1 - r0)
        other = {"precision": oname, "dtype": DTYPE[oname], "Mpx_s": round(px_rank / ok / 1e6, 1),
                 "kernel_ms": round(ok * 1e3, 4), "kernel": eng.last_kernel,
                 "frac": round((bytes_per_px(pf) + bytes_per_px(pf_out)) * px_rank / ok / 1e9 / HBM_PEAK_GBPS, 4)}
        eng.set_precision(args.precision)
        if rank == 0:
            log(f"[other precision] {other}")

    strong = None
    if world > 1 and not args.no_strong:
        # the same FRAMES frames as a 1-GPU run, rows split N ways
        sw, sk = time_steps(eng, job, src, dst, args.interp, args.steps, max(2, args.warmup // 4), world, r0, r1 - r0,
                            nframes=args.frames)
        (sw, sk), _ = reduce_max_sum(eng, world, [sw, sk], [0.0])
        strong = {"value": round(args.frames * w * h * args.steps / sw / 1e6, 1), "unit": "Mpixels/s",
                  "frames": args.frames, "rows_per_gpu": r1 - r0, "ms_per_step": round(sw / args.steps * 1e3, 4),
                  "kernel_ms": round(sk * 1e3, 4),
                  "note": "strong scaling: the N=1 workload (same frames), every frame split into N row blocks"}

    extra = {}
    if not args.no_extra and rank == 0 and world == 1 and pf.family == "yuv" and args.dither == "none":
        modes = ("tetrahedral", "trilinear") if args.extra_modes else (args.interp,)
        nf_x = min(nframes, 64)
        for dname in EXTRA_DISTS:
            if dname == args.dist:
                continue
            s2 = build_batch(eng, pf, w, h, nf_x, dname, args.unique, full_range)
            d2 = [t[:nf_x] for t in dst]
            for mode in modes:
                key = dname if len(modes) == 1 else f"{dname}/{mode}"
                extra[key] = {}
                for prec in ("strict", "fast"):
                    eng.set_precision(prec)
                    _, k2 = time_steps(eng, job, s2, d2, mode, 8, 3, 1)
                    eng.tile_stats(True)
                    job.apply(eng, s2, d2, mode)
                    st = eng.tile_stats(False)
                    extra[key][prec] = {"Mpx_s": round(nf_x * w * h / k2 / 1e6, 1), "kernel": eng.last_kernel,
                                        "tiles": st["tiles"], "tube_tiles": st["tube_tiles"], "mixed_tiles": st["mixed_tiles"],
                                        "level2_tiles": st["level2_tiles"],
                                        "window_misses": st["misses"], "gather_tiles": st["global_tiles"],
                                        "windows_staged": st["staged"]}
                    log(f"[extra] {key:14s} {prec:6s} {extra[key][prec]}")
                eng.set_precision(args.precision)
            del s2

    host_pipe = None
    if args.pipeline != "hbm" and pf.family == "yuv" and args.dither == "none":
        host_pipe = host_pipeline_leg(eng, args, job, pf, w, h, rank, world)
        if rank == 0:
            log(f"[host pipeline] {host_pipe}")

    host_cost = None
    if rank == 0 and world == 1 and pf.family == "yuv" and args.dither == "none" and not args.no_host_cost:
        host_cost = host_us_per_apply(eng, job, args)
        log(f"[host cost] {host_cost}")

    if rank == 0:
        bpp_in, bpp_out = bytes_per_px(pf), bytes_per_px(pf_out)
        bpp = bpp_in + bpp_out
        lattice_bytes = 3 * args.lut ** 3 * 4
        bytes_launch = bpp * px_rank + lattice_bytes              # per launch on one GPU
        achieved = bytes_launch / kern / 1e9
        tag = f"{args.size}_{args.fmt}_{args.interp}_lut{args.lut}_{args.dist}_f{args.frames}"
        if args.precision != "strict":
            tag += f"_{args.precision}"
        traffic = load_traffic(tag) if (pf_out is pf and args.range_src == "tv") else None
        chain = "lut3d" if pf.family != "yuv" else (
            ("scale=in_range=pc:out_range=tv,format=8-bit -> " if full_range else "") + "yuv->rgb -> lut3d -> rgb->yuv")
        result = {
            "metric": "Mpixels/s (+ achieved HBM GB/s %peak), UHD 10-bit tetrahedral, 1/2/4/8 MI355X",
            "value": round(px_total * args.steps / wall / 1e6, 1),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": DTYPE[args.precision],
            "data": f"synthetic ({args.dist}: {args.unique} seeded frames tiled to the batch; generated log709 .cube)",
            "config": {
                "workload": f"{w}x{h} {args.fmt}" + (f" -> {args.out_fmt}" if pf_out is not pf else "") +
                            f", {args.lut}^3 log->Rec.709 .cube, {args.interp}, {args.frames} frames/GPU/step resident in HBM, "
                            f"row-block shard x{world} (rows [{r0},{r1}) of shared full-height planes on rank 0)",
                "frames_per_gpu": args.frames, "lut_size": args.lut, "interp": args.interp,
                "pix_fmt": args.fmt, "out_pix_fmt": args.out_fmt or args.fmt, "range_src": args.range_src,
                "chain": chain, "precision": args.precision,
                "distribution": args.dist, "kernel": kernel_name, "lds_window": tile_stats,
                "dither": args.dither,
                "parallelism": f"row-block x{world}", "bytes_per_pixel": bpp, "setup_s": round(t_setup, 2),
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "traffic_source": ("profiles/traffic.json: committed rocprofv3 --pmc summary of this workload "
                                   "(FETCH_SIZE x2 + WRITE_SIZE), NOT measured in this run") if traffic else None,
                "kernel_ms": round(kern * 1e3, 4), "algorithmic_bytes_per_launch": int(bytes_launch),
                "read_GBps": round((bpp_in * px_rank + lattice_bytes) / kern / 1e9, 1),
            },
        }
        issue = load_traffic(tag, "issue.json") if traffic else None
        if issue:
            # what actually bounds the kernel: VALU instruction issue, from the same committed counter passes
            result["roofline"]["valu_issue"] = dict(issue, note="committed rocprofv3 --pmc summary of this workload, NOT measured in this "
                                                    "run: SQ_INSTS_VALU x 2.5 cycles / (1024 SIMDs x wave lifetime)")
        if other:
            result["other_precision"] = other
        if collective:
            result["collective"] = collective
        if strong:
            result["strong"] = strong
        if extra:
            result["extra_Mpx_s"] = extra
        if host_pipe:
            result["host_pipeline"] = host_pipe
        if host_cost:
            result["host_us_per_apply"] = host_cost
        if not args.no_cpu_baseline and world == 1 and pf.family != "packed":
            result["cpu_baseline"] = cpu_baseline(lut, job, w, h, args.interp, args.dist, args.cpu_seconds)
        elif world > 1:
            result["cpu_baseline"] = None          # timed on rank 0 at N=1 only
        print(json.dumps(result), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
