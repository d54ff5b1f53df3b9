"""Row-block partition of a frame across GPUs.

Every output pixel depends on one input pixel, its shared chroma sample and the read-only
lattice, so frames split into independent row blocks with no halo and no exchange step.
The split is FFmpeg's own slice rule -- rows [h*j/n, h*(j+1)/n) (vf_lut3d.c slice threading,
SURVEY.md 3.4) -- with boundaries rounded to the chroma block height so 4:2:0 chroma rows
never straddle two ranks.  The only collective on the path is the lattice broadcast
(`LutEngine.set_lut_distributed`).
"""
from __future__ import annotations

from typing import List, Tuple


def row_blocks(h: int, n: int, align: int = 2) -> List[Tuple[int, int]]:
    """Split rows [0,h) into `n` contiguous blocks whose starts are multiples of `align`.
    Blocks may be empty when h < n*align.  The last block ends at h even if h is odd."""
    if n < 1 or align < 1 or h < 0:
        raise ValueError("bad partition request")
    units = (h + align - 1) // align
    cuts = [min(h, (units * j // n) * align) for j in range(n)] + [h]
    return [(cuts[j], cuts[j + 1]) for j in range(n)]


def my_rows(h: int, rank: int, world: int, align: int = 2) -> Tuple[int, int]:
    return row_blocks(h, world, align)[rank]


def broadcast_lattice(lut, src: int = 0, group=None, device=None):
    """The path's only collective, in its backend-neutral form: rank `src` holds a parsed
    CubeLut, every rank returns (n, scale float32[3], table float32[n,n,n,3]).

    Two broadcasts (4 floats of metadata, then 3*n^3 floats: 431,244 B for 33^3).  Over gloo it
    runs on CPU tensors (tests); over nccl (= RCCL, xGMI) pass device=cuda:<local rank>.
    `LutEngine.set_lut_distributed` is the zero-copy GPU variant: it broadcasts straight into
    each rank's device lattice."""
    import numpy as np
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    meta = torch.zeros(4, dtype=torch.float32, device=dev)
    if rank == src:
        if lut is None:
            raise ValueError("the source rank must pass the LUT")
        meta = torch.tensor([float(lut.n), *[float(v) for v in lut.scale]], dtype=torch.float32, device=dev)
    dist.broadcast(meta, src=src, group=group)
    n = int(meta[0].item())
    if rank == src:
        table = torch.from_numpy(np.ascontiguousarray(lut.table, dtype=np.float32)).to(dev).reshape(-1)
    else:
        table = torch.empty(n * n * n * 3, dtype=torch.float32, device=dev)
    dist.broadcast(table, src=src, group=group)
    scale = meta[1:].cpu().numpy().astype(np.float32)
    return n, scale, table.cpu().numpy().reshape(n, n, n, 3)
