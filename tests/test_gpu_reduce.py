"""The multi-GPU end of a render job on the one GPU a test box has.

* pt_film_reduce over RCCL with a one-rank communicator: librccl is bound at run time, a communicator
  is created through the library's helpers, and render -> reduce -> tone map runs as one
  stream-ordered sequence (no host synchronisation in between).  With one rank the reduce is the
  identity, so the film and the display image must equal a run without it, bit for bit.
* Two processes sharing the GPU run sharding.render_job - the call bench.py makes - over gloo: rank 1
  never renders iteration 0 and depends on the reset, both ranks reduce, rank 0 ends up with the sum.
  Exact against the rank-ordered sum of the private films; against the one-process film up to the
  fp32 summation order (|d| <= 1e-6 + 1e-5 |ref|)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_film_reduce_over_rccl_with_one_rank(pt, gpu_ctx, scene_factory, devbuf):
    w, h = 160, 120
    hs = scene_factory("helmet", w, h, 8)
    gpu_ctx.upload(hs.desc())
    gpu_ctx.integrator_init(w, h)
    for k in range(2):
        gpu_ctx.render_batch(w, h, 8 * k, 8)
    gpu_ctx.synchronize()
    expected = gpu_ctx.film_download(w, h)
    expected_rgba = gpu_ctx.display_download(w, h, 16)
    uid = gpu_ctx.rccl_unique_id()
    assert len(uid) == 128 and any(uid)
    comm = gpu_ctx.rccl_comm_create(1, uid, 0)
    assert comm
    try:
        rgba = devbuf(w * h * 4)
        for k in range(2):
            gpu_ctx.render_batch(w, h, 8 * k, 8)
        gpu_ctx.film_reduce(comm, 0)           # ordered after the passes in flight, on the context's stream
        gpu_ctx.tonemap(0, w, h, 16, rgba.ptr)  # and this after the reduce
        got = gpu_ctx.film_download(w, h)
        assert np.array_equal(got.view(np.uint32), expected.view(np.uint32))
        assert np.array_equal(rgba.download(np.uint8, w * h * 4).reshape(h, w, 4), expected_rgba)
        with pytest.raises(pt.PtError, match="rank"):
            gpu_ctx.rccl_comm_create(1, uid, 1)
    finally:
        gpu_ctx.rccl_comm_destroy(comm)


WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["PT_ROOT"])
import cuda_path_tracer_amd as pt

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", init_method=f"tcp://127.0.0.1:{os.environ['MASTER_PORT']}", rank=rank, world_size=world)
w, h, total, block = 96, 72, 14, 4
hs = pt.SceneSpec.load(os.path.join(os.environ["PT_ROOT"], "scenes", "teapot.npz")).build(width=w, height=h, max_depth=6)
ctx = pt.Context(0)
hs.update_device(ctx)
ctx.integrator_init(w, h)
ctx.render_batch(w, h, 0, 3)  # leftovers of an earlier job in the film: the job's reset must clear them on every rank
ctx.synchronize()
film = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda:0")
ctx.integrator_reset(w, h)
for first, count in pt.sharding.iteration_blocks(rank, world, total, block):
    ctx.render_batch(w, h, first, count)
ctx.synchronize()
np.save(os.path.join(os.environ["PT_OUT"], f"private_rank{rank}.npy"), ctx.film_download(w, h))
done = pt.sharding.render_job(ctx, w, h, total, block, rank, world, film)
np.save(os.path.join(os.environ["PT_OUT"], f"job_rank{rank}.npy"), ctx.film_download(w, h))
with open(os.path.join(os.environ["PT_OUT"], f"done_rank{rank}.txt"), "w") as f:
    f.write(str(done))
dist.barrier()
ctx.close()
dist.destroy_process_group()
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_on_one_gpu_run_the_bench_render_job(pt, gpu_ctx, scene_factory, tmp_path):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PT_ROOT=ROOT, PT_OUT=str(tmp_path))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=200)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode())
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert [int(open(tmp_path / f"done_rank{r}.txt").read()) for r in range(2)] == [8, 6]  # blocks 0,2 (4+4) and 1,3 (4+2)
    priv = [np.load(tmp_path / f"private_rank{r}.npy") for r in range(2)]
    job = [np.load(tmp_path / f"job_rank{r}.npy") for r in range(2)]
    assert np.array_equal(job[1].view(np.uint32), priv[1].view(np.uint32))  # ranks other than the root keep their own film
    assert np.array_equal(job[0].view(np.uint32), (priv[0] + priv[1]).view(np.uint32))
    w, h = 96, 72
    hs = scene_factory("teapot", w, h, 6)
    gpu_ctx.upload(hs.desc())
    gpu_ctx.integrator_init(w, h)
    gpu_ctx.render_batch(w, h, 0, 14)
    single = gpu_ctx.film_download(w, h)
    assert np.all(np.abs(job[0] - single) <= 1e-6 + 1e-5 * np.abs(single)) and single.mean() > 1e-3
