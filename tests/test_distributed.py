"""The N > 1 path on CPU: two gloo ranks deal out iteration blocks (sharding.iteration_blocks),
render their own blocks, and sum the films with one reduce (sharding.reduce_film).  The renderer
in these workers is the CPU oracle (there is no GPU here); what is under test is the sharding
plan and the reduce, which are the same code bench.py runs over RCCL."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["PT_ROOT"]); sys.path.insert(0, os.path.join(os.environ["PT_ROOT"], "tests"))
import cuda_path_tracer_amd as pt
import oracle_lib as ora
from conftest import make_tiny_scene

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", init_method=f"tcp://127.0.0.1:{os.environ['MASTER_PORT']}", rank=rank, world_size=world)
hs = make_tiny_scene(pt)
desc = hs.desc()
w, h, total, block = 32, 24, 7, 2
cam = pt.camera_make([0.0, 2.0, 6.0], [0.0, 1.0, 0.0], [0.0, 1.0, 0.0], 50.0, w, h)
desc.camera = cam
film = np.zeros((h, w, 3), np.float32)
mine = []
if os.environ.get("PT_SHARD", "samples") == "samples":
    for first, count in pt.sharding.iteration_blocks(rank, world, total, block):
        for it in range(first, first + count):
            film, _ = ora.render(desc, w, h, it, film, nthreads=2)
            mine.append(it)
else:  # tiles: every iteration of this rank's band of rows (the oracle renders whole frames; the band is cut out)
    x0, y0, ww, hh = pt.sharding.tile_windows(world, w, h)[rank]
    for it in range(total):
        film, _ = ora.render(desc, w, h, it, film, nthreads=2)
        mine.append(it)
    band = np.zeros_like(film)
    band[y0:y0 + hh, x0:x0 + ww] = film[y0:y0 + hh, x0:x0 + ww]
    film = band
np.save(os.path.join(os.environ["PT_OUT"], f"film_rank{rank}.npy"), film)
t = torch.from_numpy(film.copy())
pt.sharding.reduce_film(t, dst=0)
slow = pt.sharding.max_over_ranks(float(rank + 1), "cpu")
assert slow == float(world)
if rank == 0:
    np.save(os.path.join(os.environ["PT_OUT"], "film_reduced.npy"), t.numpy())
with open(os.path.join(os.environ["PT_OUT"], f"iters_rank{rank}.txt"), "w") as f:
    f.write(" ".join(map(str, mine)))
dist.barrier()
dist.destroy_process_group()
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_iteration_blocks_partition_the_job(pt):
    for world in (1, 2, 3, 8):
        for total, block in ((0, 4), (1, 1), (7, 2), (32, 8), (33, 8), (5, 16)):
            owned = sorted(pt.sharding.all_owned(world, total, block))
            assert owned == list(range(total))
            for r in range(world):
                for first, count in pt.sharding.iteration_blocks(r, world, total, block):
                    assert first % block == 0 and 1 <= count <= block and (first // block) % world == r
    with pytest.raises(ValueError):
        pt.sharding.iteration_blocks(2, 2, 8, 1)
    with pytest.raises(ValueError):
        pt.sharding.iteration_blocks(0, 1, 8, 0)


def test_rays_shard_covers_batch(pt):
    for world in (1, 2, 3, 8):
        for n in (0, 1, 7, 64, 4960689):
            cuts = [pt.sharding.rays_shard(r, world, n) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_tile_windows_partition_the_frame(pt):
    for world in (1, 2, 3, 8, 16):
        for width, height in ((16, 10), (1920, 1080), (3840, 2160), (7, 5), (64, 64)):
            wins = pt.sharding.tile_windows(world, width, height)
            assert len(wins) == world
            covered = np.zeros((height, width), np.int32)
            for x0, y0, w, h in wins:
                assert x0 == 0 and w == width and h >= 0 and y0 % 8 == 0 or h == 0
                covered[y0:y0 + h, x0:x0 + w] += 1
            assert (covered == 1).all()
            heights = [h for _, _, _, h in wins]
            if height >= 8 * world:
                assert max(heights) - min(heights) <= 8
    with pytest.raises(ValueError):
        pt.sharding.tile_windows(0, 8, 8)


def _run_two_ranks(tmp_path, shard):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PT_ROOT=ROOT, PT_OUT=str(tmp_path), OMP_NUM_THREADS="2", PT_SHARD=shard)
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode())
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)


def test_two_rank_gloo_tile_sharded_film_reduce(pt, ora, tiny_scene, tmp_path):
    """Image-tile plan: each rank holds every iteration of its own band of rows and zeros elsewhere, so the
    reduced film equals the single-process film bit for bit (every pixel has one non-zero addend)."""
    _run_two_ranks(tmp_path, "tiles")
    w, h = 32, 24
    f0, f1 = np.load(tmp_path / "film_rank0.npy"), np.load(tmp_path / "film_rank1.npy")
    assert (f0[16:] == 0).all() and (f1[:16] == 0).all() and f0[:16].any() and f1[16:].any()
    reduced = np.load(tmp_path / "film_reduced.npy")
    desc = tiny_scene.desc()
    desc.camera = pt.camera_make([0.0, 2.0, 6.0], [0.0, 1.0, 0.0], [0.0, 1.0, 0.0], 50.0, w, h)
    film = np.zeros((h, w, 3), np.float32)
    for it in range(7):
        film, _ = ora.render(desc, w, h, it, film, nthreads=4)
    assert np.array_equal(reduced.view(np.uint32), film.view(np.uint32))


def test_two_rank_gloo_film_reduce(pt, ora, tiny_scene, tmp_path):
    _run_two_ranks(tmp_path, "samples")

    iters = [list(map(int, open(tmp_path / f"iters_rank{r}.txt").read().split())) for r in range(2)]
    assert iters[0] == [0, 1, 4, 5] and iters[1] == [2, 3, 6]

    # the reduce is the rank-ordered sum of the private films, exactly (two addends: order-free)
    f0, f1 = np.load(tmp_path / "film_rank0.npy"), np.load(tmp_path / "film_rank1.npy")
    reduced = np.load(tmp_path / "film_reduced.npy")
    assert np.array_equal((f0 + f1).view(np.uint32), reduced.view(np.uint32))

    # and equals the single-process film of the same 7 iterations up to fp32 summation order
    w, h = 32, 24
    desc = tiny_scene.desc()
    desc.camera = pt.camera_make([0.0, 2.0, 6.0], [0.0, 1.0, 0.0], [0.0, 1.0, 0.0], 50.0, w, h)
    film = np.zeros((h, w, 3), np.float32)
    for it in range(7):
        film, _ = ora.render(desc, w, h, it, film, nthreads=4)
    assert np.allclose(reduced, film, rtol=1e-5, atol=1e-6)
    assert film.mean() > 1e-3
