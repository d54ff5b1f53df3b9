"""Row-block partition and the N>1 control flow on CPU with gloo, world_size 2.

The data path has no collective: ranks own disjoint row blocks.  The only collective is the
lattice broadcast at LUT load; here it runs over gloo on CPU tensors through the same helper
bench.py uses on GPUs (`shard.broadcast_lattice`), and each rank's block of an oracle-applied
frame is checked against the whole-frame result.
"""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

from lut_renderer_amd.shard import my_rows, row_blocks

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("h,n,align", [(2160, 8, 2), (2160, 7, 2), (1080, 8, 2), (4320, 8, 2), (17, 4, 2), (2, 8, 2),
                                       (0, 3, 2), (1081, 5, 2), (270, 8, 1)])
def test_row_blocks_partition(h, n, align):
    blocks = row_blocks(h, n, align)
    assert len(blocks) == n
    assert blocks[0][0] == 0 and blocks[-1][1] == h
    for (a0, a1), (b0, b1) in zip(blocks, blocks[1:]):
        assert a1 == b0 and a0 <= a1
    for a0, _ in blocks:
        assert a0 % align == 0 or a0 == h
    sizes = [b - a for a, b in blocks if b > a]
    if h >= n * align:
        assert max(sizes) - min(sizes) <= align        # FFmpeg's h*j/n slicing is balanced
    assert [my_rows(h, r, n, align) for r in range(n)] == blocks


def test_uhd_over_8_gpus_is_270_rows_each():
    assert row_blocks(2160, 8, 2) == [(270 * i, 270 * (i + 1)) for i in range(8)]
    assert row_blocks(4320, 8, 2) == [(540 * i, 540 * (i + 1)) for i in range(8)]


def test_bad_requests():
    for args in ((10, 0, 2), (10, 2, 0), (-1, 2, 2)):
        with pytest.raises(ValueError):
            row_blocks(*args)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmp):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from lut_renderer_amd import cube, frames
    from lut_renderer_amd.shard import broadcast_lattice, my_rows
    from oracle import binding as orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank 0 parses the cube; everyone else learns n / scale / lattice from ONE broadcast pair
        lut = cube.read_cube(Path(tmp) / "log709_17.cube") if rank == 0 else None
        n, scale, table = broadcast_lattice(lut, src=0)
        assert n == 17 and table.shape == (17, 17, 17, 3)
        ref = cube.read_cube(Path(tmp) / "log709_17.cube")
        assert np.array_equal(table, ref.table) and np.array_equal(scale, ref.scale)
        # every rank applies its own row block; blocks are disjoint and need no exchange
        w, h = 64, 36
        src = frames.natural_yuv(w, h, 10, 1, 1, k=4)
        r0, r1 = my_rows(h, rank, world, align=2)
        k = orc.yuv_constants(din=10)
        mine = orc.apply_yuv(table, scale, "tetrahedral", k, 10, 10, 10, 1, 1,
                             [src[0][r0:r1], src[1][r0 // 2:r1 // 2], src[2][r0 // 2:r1 // 2]])
        whole = orc.apply_yuv(table, scale, "tetrahedral", k, 10, 10, 10, 1, 1, src)
        assert np.array_equal(mine[0], whole[0][r0:r1])
        assert np.array_equal(mine[1], whole[1][r0 // 2:r1 // 2])
        # the max-over-ranks timing reduction bench.py performs
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == world
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_broadcast_and_disjoint_blocks(tmp_path):
    import torch.multiprocessing as mp
    from lut_renderer_amd import cube
    cube.write_cube(tmp_path / "log709_17.cube", cube.log709_lattice(17))
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
