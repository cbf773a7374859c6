"""Screen partition across GPUs and the gather of the float4 pieces.

The image is cut into 8-row strips dealt round-robin to ranks (strip s belongs to rank s % world), so
light-facing and sky rows are spread evenly.  Every pixel's RNG stream depends only on (x, y, frame)
(main.glsl:176-181,409), so the assembled image is bit-identical to the single-GPU one.  The only
exchange is one gather of each rank's float4 piece to rank 0 at the end of a render; the device-side
mirror of `assemble` is jpt_assemble_from_ranks (csrc/jpt_kernels_post.hip).
"""
from __future__ import annotations

import numpy as np

STRIP_ROWS = 8  # == jpt::kStripRows


def rows_of_rank(height: int, rank: int, world: int) -> np.ndarray:
    """Image rows rendered by `rank`, in the order of its local framebuffer."""
    n_strips = (height + STRIP_ROWS - 1) // STRIP_ROWS
    rows = []
    for s in range(rank, n_strips, world):
        rows.extend(range(s * STRIP_ROWS, min((s + 1) * STRIP_ROWS, height)))
    return np.asarray(rows, dtype=np.int64)


def max_local_rows(height: int, world: int) -> int:
    return max(len(rows_of_rank(height, r, world)) for r in range(world))


def extract_piece(image: np.ndarray, rank: int, world: int) -> np.ndarray:
    """The rank's piece of a full [H, W, C] image, padded to the common piece size."""
    h = image.shape[0]
    rows = rows_of_rank(h, rank, world)
    piece = np.zeros((max_local_rows(h, world),) + image.shape[1:], dtype=image.dtype)
    piece[: len(rows)] = image[rows]
    return piece


def assemble(pieces, height: int, world: int) -> np.ndarray:
    """Rank-major pieces [world, max_local_rows, W, C] -> full image [H, W, C]."""
    pieces = np.asarray(pieces)
    out = np.zeros((height,) + pieces.shape[2:], dtype=pieces.dtype)
    for r in range(world):
        rows = rows_of_rank(height, r, world)
        out[rows] = pieces[r, : len(rows)]
    return out


def gather_to_rank0(piece, dist, rank: int, world: int, gathered=None):
    """One exchange per render: every rank's piece to rank 0 (torch tensors; RCCL on GPUs, gloo on CPU).
    Each peer's piece travels point-to-point to rank 0 -- on MI355X that is one xGMI link per peer.  There is no
    fallback: a backend that cannot gather raises (an all_gather in its place would put 8x the bytes on every link
    without anybody noticing)."""
    import torch
    if world == 1 and (dist is None or not dist.is_initialized()):
        return piece.unsqueeze(0)
    # (a process group of ONE rank still goes through the backend: that is how the GPU box's single device rehearses the RCCL call)
    if rank == 0 and gathered is None:
        gathered = torch.empty((world,) + tuple(piece.shape), dtype=piece.dtype, device=piece.device)
    dist.gather(piece, list(gathered.unbind(0)) if rank == 0 else None, dst=0)
    return gathered if rank == 0 else None
