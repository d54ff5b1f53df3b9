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
