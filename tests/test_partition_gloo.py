"""Multi-GPU path on CPU: 2 processes, gloo backend.  Each rank 'renders' its strips (with the oracle,
which stands in for the device kernels here), the float4 pieces and the rgba8 display rows are gathered to
rank 0 with the same helper bench.py uses, and the assembled image must be bit-identical to the single-process image."""
import os
import socket
import sys

import numpy as np
import pytest

from gdpathtracing_amd import partition

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("h,world", [(1080, 8), (1080, 1), (45, 2), (7, 3), (64, 4), (2160, 8), (9, 2)])
def test_strip_partition_covers_every_row_once(h, world):
    rows = np.concatenate([partition.rows_of_rank(h, r, world) for r in range(world)])
    assert sorted(rows.tolist()) == list(range(h))
    sizes = [len(partition.rows_of_rank(h, r, world)) for r in range(world)]
    assert max(sizes) - min(sizes) <= partition.STRIP_ROWS
    assert partition.max_local_rows(h, world) == max(sizes)
    # strip s belongs to rank s % world (mirrors jpt::local_to_global_row)
    for r in range(world):
        rr = partition.rows_of_rank(h, r, world)
        assert ((rr // partition.STRIP_ROWS) % world == r).all()


def test_extract_and_assemble_roundtrip():
    rng = np.random.RandomState(0)
    img = rng.rand(45, 16, 4).astype(np.float32)
    for world in (1, 2, 3, 8):
        pieces = np.stack([partition.extract_piece(img, r, world) for r in range(world)])
        assert np.array_equal(partition.assemble(pieces, 45, world), img)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from gdpathtracing_amd import partition, scenes, wire
    from oracle import binding as ob
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = scenes.cornell_scene()
        w, h = 40, 45
        cam = scenes.camera_block(sc.camera, w, h)
        ref = ob.build_scene(sc)
        # the oracle renders whole images; a rank keeps only its strips (per-pixel RNG streams make the
        # rows independent, main.glsl:176-181)
        full, full_ldr, _, _, _ = ob.render(ref, cam, w, h, 2, 2, 1, wire.ACCUM_REF_LDR8, n_threads=1)
        piece = torch.from_numpy(partition.extract_piece(full, rank, world))
        if rank != 0:
            piece = piece.clone()
        g = partition.gather_to_rank0(piece, dist, rank, world)
        # what bench.py gathers by default: each rank's finished rgba8 rows, as one int32 per pixel
        ldr_piece = torch.from_numpy(partition.extract_piece(full_ldr, rank, world).view(np.int32).copy())
        gl = partition.gather_to_rank0(ldr_piece, dist, rank, world)
        if rank == 0:
            img = partition.assemble(g.numpy(), h, world)
            ldr = partition.assemble(gl.numpy().view(np.uint8).reshape(world, -1, w, 4), h, world)
            q.put(("ok", bool(np.array_equal(img, full)) and bool(np.array_equal(ldr, full_ldr)), float(np.abs(img).sum())))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_process_gloo_gather_is_bit_identical():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    tag, same, total = q.get(timeout=10)
    assert tag == "ok" and same and total > 0
