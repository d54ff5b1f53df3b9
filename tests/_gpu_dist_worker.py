"""Worker of tests/test_gpu_parity.py::test_two_rank_lut_broadcast_and_row_blocks (started by torchrun).

Every rank of the job does what bench.py / a multi-GPU caller does: rank 0 parses the cube, all ranks call
`LutEngine.set_lut_distributed` (the non-source ranks allocate, receive the broadcast into their device
lattice and seal it), then each rank applies its own row block and saves it.  On the one-GPU test box both
ranks share cuda:0 and the collective runs over gloo; on a multi-GPU node the same code runs over RCCL.
"""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from lut_renderer_amd import cube, frames  # noqa: E402
from lut_renderer_amd.engine import LutEngine  # noqa: E402
from lut_renderer_amd.shard import my_rows  # noqa: E402


def main():
    out_dir, cube_path = Path(sys.argv[1]), Path(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = int(os.environ.get("LUTR_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    dist.init_process_group(os.environ.get("LUTR_DIST_BACKEND", "nccl"))
    try:
        with LutEngine(dev) as eng:
            eng.set_lut_distributed(cube.read_cube(cube_path) if rank == 0 else None, src=0)
            w, h = 256, 72
            src = frames.natural_yuv(w, h, 10, 1, 1, k=9)
            r0, r1 = my_rows(h, rank, world, align=2)
            planes = [torch.from_numpy(p.view(np.int16)).to(eng.device) for p in src]
            dst = [torch.zeros_like(t) for t in planes]
            eng.apply_yuv(planes, dst, pix_fmt="yuv420p10le", row0=r0, rows=r1 - r0)
            torch.cuda.synchronize()
            np.savez(out_dir / f"rank{rank}.npz", r0=r0, r1=r1, kernel=eng.last_kernel,
                     y=dst[0].cpu().numpy().view(np.uint16), cb=dst[1].cpu().numpy().view(np.uint16),
                     cr=dst[2].cpu().numpy().view(np.uint16))
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
