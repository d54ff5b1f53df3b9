#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the 3D-LUT apply hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the fused kernel over a batch of synthetic frames that is already
resident in HBM.  Default workload = BASELINE.json configs[1]: 3840x2160 yuv420p10le,
33^3 log->Rec.709 .cube, tetrahedral, 1 GPU, 256 frames per launch (12.7 GB of the 288 GB HBM).  With N > 1 every frame is split into N row
blocks; rank g owns block g of N x FRAMES frames, so per-GPU work is fixed
("weak" scaling); the only collective is the RCCL broadcast of the lattice at LUT load.

Rank 0 prints ONE JSON line.  `roofline.achieved` = algorithmic bytes per launch (6 B/px
for yuv420p10le in+out, + the 431,244 B lattice) / the kernel's mean launch duration
measured with HIP events on the launch stream.  `cpu_baseline` times the CPU oracle
(kind "port": a restatement of FFmpeg lut3d, not FFmpeg) on the host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

SIZES = {"1080p": (1920, 1080), "uhd": (3840, 2160), "8k": (7680, 4320)}
HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--frames", type=int, default=256,
                    help="frames per GPU per step (SURVEY 8d: >= 64; 256 = the batch of BASELINE config 5: 6.4 GB in + 6.4 GB out)")
    ap.add_argument("--size", default="uhd", choices=sorted(SIZES))
    ap.add_argument("--fmt", default="yuv420p10le")
    ap.add_argument("--interp", default="tetrahedral")
    ap.add_argument("--lut", type=int, default=33, help="lattice size N of the generated log709 LUT")
    ap.add_argument("--dist", default="natural", choices=["natural", "uniform"])
    ap.add_argument("--variant", default="auto")
    ap.add_argument("--unique", type=int, default=2, help="distinct synthetic frames tiled into the batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stats", action="store_true", help="skip the extra LDS-window statistics pass (profiling runs)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0,
                    help="wall budget of the CPU baseline sample (all host cores: ~10 s wall on 256 threads)")
    ap.add_argument("--extra", action="store_true", help="also time the other distribution / mode (stderr only)")
    ap.add_argument("--prewarm-ms", type=float, default=60.0,
                    help="untimed kernel launches before the W warm-up steps until this much time has passed: the GPU "
                         "needs ~20 ms of load to leave its idle clock (DESIGN.md 5), whatever W the caller picks")
    ap.add_argument("--dither", default="none", choices=["none", "error_diffusion"],
                    help="also dither the final quantisation (reference option zscale_dither; YUV formats, informational)")
    ap.add_argument("--pipeline", default="hbm", choices=["hbm", "host"],
                    help="host: also time BASELINE config 5 (frames in pinned host memory, overlapped copies); "
                         "reported as `host_pipeline`, never as `value`")
    ap.add_argument("--host-frames", type=int, default=256)
    return ap.parse_args()


def dist_setup(gpus: int):
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
    elif gpus != 1:
        raise SystemExit("for --gpus N > 1 launch with torch.distributed.run (one rank per GPU)")
    return rank, local, world


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


DITHER = "none"


def apply(eng, pf, src, dst, fmt, interp):
    if DITHER != "none":
        return eng.apply_yuv(src, dst, pix_fmt=fmt, interp=interp, dither=DITHER)
    if pf.family == "packed":
        return eng.apply_packed(src[0], dst[0], pix_fmt=fmt, interp=interp)
    if pf.family == "gbr":
        return eng.apply_rgb(src, dst, depth=pf.depth, interp=interp)
    return eng.apply_yuv(src, dst, pix_fmt=fmt, interp=interp)


def build_batch(eng, pf, w, h, r0, r1, nframes, dist_name, unique):
    """Rows [r0,r1) of `unique` synthetic frames, tiled to `nframes` frames on the device."""
    from lut_renderer_amd import frames
    if pf.family == "packed":                            # interleave the planar RGB generator's frames
        imgs = []
        for k in range(unique):
            g, b, r = (frames.natural_rgb if dist_name == "natural" else frames.uniform_rgb)(w, h, pf.depth, k=k)
            img = np.full((r1 - r0, w, pf.nc), (1 << pf.depth) - 1, dtype=g.dtype)
            img[..., pf.rgb[0]], img[..., pf.rgb[1]], img[..., pf.rgb[2]] = r[r0:r1], g[r0:r1], b[r0:r1]
            imgs.append(torch.from_numpy(img.view(np.int16) if img.dtype == np.uint16 else img))
        u = torch.stack(imgs).to(eng.device)
        return [u.repeat((nframes + unique - 1) // unique, 1, 1, 1)[:nframes].contiguous()]
    bh = 1 << pf.csy
    planes = [[], [], []]
    for k in range(unique):
        if pf.family == "gbr":
            f = (frames.natural_rgb if dist_name == "natural" else frames.uniform_rgb)(w, h, pf.depth, k=k)
        else:
            f = frames.make_yuv(dist_name, w, h, pf.depth, pf.csx, pf.csy, k=k)
        sl = [f[0][r0:r1], f[1][r0 // bh:(r1 + bh - 1) // bh], f[2][r0 // bh:(r1 + bh - 1) // bh]]
        for i in range(3):
            a = np.ascontiguousarray(sl[i])
            planes[i].append(torch.from_numpy(a.view(np.int16) if a.dtype == np.uint16 else a))
    out = []
    for i in range(3):
        u = torch.stack(planes[i]).to(eng.device)
        reps = (nframes + unique - 1) // unique
        out.append(u.repeat(reps, 1, 1)[:nframes].contiguous())
    return out


PREWARM_MS = 0.0


def time_steps(eng, pf, src, dst, fmt, interp, steps, warmup, world):
    t_end = time.perf_counter() + PREWARM_MS / 1e3
    while time.perf_counter() < t_end:                 # clock ramp; not part of W, not timed
        apply(eng, pf, src, dst, fmt, interp)
        torch.cuda.synchronize()
    for _ in range(warmup):
        apply(eng, pf, src, dst, fmt, interp)
    torch.cuda.synchronize()
    barrier(world)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        apply(eng, pf, src, dst, fmt, interp)
    ev1.record()
    torch.cuda.synchronize()
    barrier(world)
    wall = time.perf_counter() - t0
    return wall, ev0.elapsed_time(ev1) / 1e3 / steps     # wall seconds, mean kernel seconds (HIP events)


def cpu_baseline(lut, pf, w, h, interp, dist_name, budget_s):
    """CPU oracle (port of FFmpeg lut3d + this repo's YUV contract) on all host cores, row-sliced
    like FFmpeg's slice threads; bounded sample of the same workload."""
    from lut_renderer_amd import frames
    from oracle import binding as orc
    cores = os.cpu_count() or 1
    # whole frames until the budget is spent (>= 2 repetitions); rows are split over `cores` threads
    if pf.family == "gbr":
        f = (frames.natural_rgb if dist_name == "natural" else frames.uniform_rgb)(w, h, pf.depth, k=0)

        def run():
            orc.apply_rgb(lut.table, lut.scale, pf.depth, interp, f, nthreads=cores)
    else:
        f = frames.make_yuv(dist_name, w, h, pf.depth, pf.csx, pf.csy, k=0)
        k = orc.yuv_constants("bt709", "tv", "bt709", "tv", pf.depth, pf.depth, pf.depth, 1 << (pf.csx + pf.csy))

        def run():
            orc.apply_yuv(lut.table, lut.scale, interp, k, pf.depth, pf.depth, pf.depth, pf.csx, pf.csy, f,
                          nthreads=cores)
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


def load_traffic(tag):
    """HBM bytes per launch from the committed rocprofv3 --pmc summary, if one exists for this workload."""
    p = ROOT / "profiles" / "traffic.json"
    if p.exists():
        try:
            return json.loads(p.read_text()).get(tag)
        except Exception:
            return None
    return None


def main():
    args = parse_args()
    global DITHER, PREWARM_MS
    DITHER = args.dither
    PREWARM_MS = args.prewarm_ms
    rank, local, world = dist_setup(args.gpus)
    from lut_renderer_amd import cube
    from lut_renderer_amd.engine import LutEngine, parse_pix_fmt
    from lut_renderer_amd.shard import my_rows

    w, h = SIZES[args.size]
    from lut_renderer_amd._native import PACKED_FORMATS
    if args.fmt in PACKED_FORMATS:
        from types import SimpleNamespace
        bits, nc, *rgb = PACKED_FORMATS[args.fmt]
        pf = SimpleNamespace(family="packed", depth=bits, csx=0, csy=0, nc=nc, rgb=rgb)
    else:
        pf = parse_pix_fmt(args.fmt)
    eng = LutEngine(local)
    eng.set_variant(args.variant)

    # LUT: generated log->Rec.709 lattice written as a real .cube, parsed by liblutr on rank 0,
    # broadcast to the other ranks (the path's only collective)
    lut = None
    if rank == 0:
        with tempfile.TemporaryDirectory() as d:
            lut = cube.read_cube(cube.write_cube(Path(d) / f"log709_{args.lut}.cube", cube.log709_lattice(args.lut),
                                                 title=f"log709 {args.lut}"))
    if world > 1:
        eng.set_lut_distributed(lut, src=0)
    else:
        eng.set_lut(lut)

    r0, r1 = my_rows(h, rank, world, align=1 << pf.csy)
    nframes = args.frames * world                     # weak scaling: block g of world x FRAMES frames
    src = build_batch(eng, pf, w, h, r0, r1, nframes, args.dist, args.unique)
    dst = [torch.empty_like(t) for t in src]
    px_rank = (r1 - r0) * w * nframes

    wall, kern = time_steps(eng, pf, src, dst, args.fmt, args.interp, args.steps, args.warmup, world)
    kernel_name = eng.last_kernel
    tile_stats = None
    if "tile" in kernel_name and not args.no_stats:   # one extra, untimed pass with the window counters armed
        eng.tile_stats(True)
        apply(eng, pf, src, dst, args.fmt, args.interp)
        tile_stats = eng.tile_stats(False)
        if rank == 0:
            log(f"[tile stats] {tile_stats}")
    t = torch.tensor([wall, kern, float(px_rank)], dtype=torch.float64, device=eng.device)
    if world > 1:
        import torch.distributed as dist
        tm = t.clone()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        ts = t.clone()
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
        wall, kern, px_total = tm[0].item(), tm[1].item(), ts[2].item()
    else:
        px_total = float(px_rank)

    extra = {}
    if args.extra and rank == 0 and world == 1:
        for dname in ("natural", "uniform"):
            for mode in ("tetrahedral", "trilinear"):
                s2 = build_batch(eng, pf, w, h, r0, r1, nframes, dname, args.unique)
                _, k2 = time_steps(eng, pf, s2, dst, args.fmt, mode, max(3, args.steps // 2), max(2, args.warmup // 2), 1)
                extra[f"{dname}/{mode}"] = round(px_rank / k2 / 1e6, 1)
                eng.tile_stats(True)
                apply(eng, pf, s2, dst, args.fmt, mode)
                log(f"[extra] {dname:8s} {mode:12s} {extra[f'{dname}/{mode}']:>12.1f} Mpx/s  ({eng.last_kernel}) "
                    f"{eng.tile_stats(False)}")
                del s2

    host_pipe = None
    if args.pipeline == "host" and pf.family == "yuv":
        # BASELINE config 5: frames queued in pinned host memory, round-robin over the GPUs (whole frames per rank,
        # every rank drives its own 3-slot ring); total = all ranks' frames / the slowest rank's time
        from lut_renderer_amd import frames as _frames
        from lut_renderer_amd.stream import HostPipeline
        pipe = HostPipeline(eng, args.fmt, w, h, batch=8, slots=3, interp=args.interp)
        full = _frames.make_yuv(args.dist, w, h, pf.depth, pf.csx, pf.csy, k=0)
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
        tot = torch.tensor([float(n_done)], dtype=torch.float64, device=eng.device)
        if world > 1:
            import torch.distributed as dist
            te = torch.tensor([el], dtype=torch.float64, device=eng.device)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            el = te.item()
        n_all = int(tot.item())
        gb = n_all * (pipe.fin.frame_bytes + pipe.fout.frame_bytes) / 1e9
        host_pipe = {"frames": n_all, "fps": round(n_all / el, 1), "Mpixels_s": round(n_all * w * h / el / 1e6, 1),
                     "pcie_GBps_each_way": round(gb / 2 / el, 1), "gpus": world,
                     "note": "frames in pinned host memory, round-robin over the GPUs, per GPU a 3-slot ring of 8-frame "
                             "batches with H2D / kernel / D2H on separate streams; PCIe Gen5 x16-bound (63 GB/s per "
                             "direction and GPU by spec); pcie_GBps_each_way is the sum over GPUs"}
        if rank == 0:
            log(f"[host pipeline] {host_pipe}")
    if rank == 0:
        bpp_in = (float(pf.nc) if pf.family == "packed" else 3.0 if pf.family == "gbr"
                  else 1.0 + 2.0 / (1 << (pf.csx + pf.csy))) * (1 if pf.depth <= 8 else 2)
        bpp = 2.0 * bpp_in                                        # in + out, same format
        lattice_bytes = 3 * args.lut ** 3 * 4
        bytes_launch = bpp * px_rank + lattice_bytes              # per launch on one GPU
        achieved = bytes_launch / kern / 1e9
        tag = f"{args.size}_{args.fmt}_{args.interp}_lut{args.lut}_{args.dist}_f{args.frames}"
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
            "dtype": "f32",
            "data": f"synthetic ({args.dist}: {args.unique} seeded frames tiled to the batch; generated log709 .cube)",
            "config": {
                "workload": f"{w}x{h} {args.fmt}, {args.lut}^3 log->Rec.709 .cube, {args.interp}, "
                            f"{args.frames} frames/GPU/step resident in HBM, row-block shard x{world}",
                "frames_per_gpu": args.frames, "lut_size": args.lut, "interp": args.interp,
                "pix_fmt": args.fmt, "distribution": args.dist, "kernel": kernel_name, "lds_window": tile_stats,
                "dither": args.dither,
                "parallelism": f"row-block x{world}", "bytes_per_pixel": bpp,
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": load_traffic(tag),
                "kernel_ms": round(kern * 1e3, 4), "algorithmic_bytes_per_launch": int(bytes_launch),
                "read_GBps": round((bpp_in * px_rank + lattice_bytes) / kern / 1e9, 1),
            },
        }
        if extra:
            result["extra_Mpx_s"] = extra
        if host_pipe:
            result["host_pipeline"] = host_pipe
        if not args.no_cpu_baseline and world == 1 and pf.family != "packed":
            result["cpu_baseline"] = cpu_baseline(lut, pf, w, h, args.interp, args.dist, args.cpu_seconds)
        elif world > 1:
            result["cpu_baseline"] = None          # timed on rank 0 at N=1 only
        print(json.dumps(result), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
