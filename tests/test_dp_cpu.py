"""Data-parallel gradient exchange on CPU: bucket planning properties and a world_size=2 gloo run of
the same GradReducer the GPU path uses (tf2_yolo_amd/dp.py)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tf2_yolo_amd import dp


def _segments(sizes):
    """contiguous segments laid out in construction order; returned in backward (reverse) order"""
    offs = np.concatenate([[0], np.cumsum(sizes)])
    segs = [(int(offs[i]), int(sizes[i])) for i in range(len(sizes))]
    return segs[::-1], int(offs[-1])


@pytest.mark.parametrize("bucket", [1, 100, 1000, 10 ** 9])
def test_plan_buckets_partition_and_order(bucket):
    rng = np.random.default_rng(0)
    sizes = (rng.integers(1, 40, 57) * 64).tolist()
    segs, total = _segments(sizes)
    buckets, closes = dp.plan_buckets(segs, total, bucket)
    # buckets tile [0,total) exactly, from the end of the buffer towards the start
    assert buckets[0][1] == total and buckets[-1][0] == 0
    for (lo, hi), (lo2, hi2) in zip(buckets, buckets[1:]):
        assert lo == hi2 and lo2 < hi2
    # a bucket is closed by the segment that owns its lowest offset, and only once
    flat = [b for c in closes for b in c]
    assert sorted(flat) == list(range(len(buckets)))
    for i, c in enumerate(closes):
        for b in c:
            assert buckets[b][0] <= segs[i][0]
    if bucket == 10 ** 9:
        assert len(buckets) == 1
    # a smaller first bucket: same tiling rules, the first bucket closes no later than before
    b2, c2 = dp.plan_buckets(segs, total, bucket, max(bucket // 4, 1))
    assert b2[0][1] == total and b2[-1][0] == 0 and b2[0][0] >= buckets[0][0]
    assert sorted(b for c in c2 for b in c) == list(range(len(b2)))


def _worker(rank, world, port, sizes, bucket_bytes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    segs, total = _segments(sizes)
    g = torch.arange(total, dtype=torch.float32) * (rank + 1)
    red = dp.GradReducer(g, segs, bucket_bytes=bucket_bytes)
    for i in range(len(segs)):          # backward order
        red.segment_done(i)
    scale = red.finish()
    params = torch.full((8,), float(rank))
    dp.broadcast_parameters([params], 0)
    q.put((rank, (g * scale).numpy(), scale, params.numpy(), len(red.buckets)))
    dist.destroy_process_group()


def test_gloo_world2_allreduce_mean():
    world = 2
    sizes = [64, 128, 64, 256, 64, 64]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, sizes, 4 * 200, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = sum(sizes)
    expect = np.arange(total, dtype=np.float32) * (1 + 2) / 2       # mean over ranks
    for rank, g, scale, params, nb in res:
        assert scale == 0.5 and nb > 1
        assert np.allclose(g, expect)
        assert np.all(params == 0.0)                                  # rank 0's weights everywhere


def test_bench_launcher_propagates_a_failing_rank():
    """`python bench.py --gpus 2` without a torchrun environment starts its own ranks (bench.launch_ranks); here there is no
    GPU, so both ranks fail at their first device call: the launcher must stop, exit non-zero and print no JSON line. (The
    working case -- two gloo ranks sharing one GPU -- is tests/test_gpu_dp.py.)"""
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: the ranks would run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, YOLO_BENCH_SINGLE_DEVICE="1", YOLO_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline", "--no-extra-blocks"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "stopping the other ranks" in r.stderr or "exited with" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
