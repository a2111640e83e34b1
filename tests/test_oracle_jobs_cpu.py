"""The worker-process path of the full-batch headline tests (tests/oracle_jobs.py, started by conftest.py in GPU sessions)
on a tiny YOLOv3 without a GPU: the job runs in a spawned worker exactly as in the GPU suite, returns NumPy only, and gives
what the same call gives in this process."""
import multiprocessing as mp
from concurrent.futures import ProcessPoolExecutor

import numpy as np

import oracle_jobs

A9 = [[0.89663461, 0.78365384], [0.375, 0.47596153], [0.27884615, 0.21634615], [0.14182692, 0.28605769],
      [0.14903846, 0.10817307], [0.07211538, 0.14663461], [0.07932692, 0.05528846], [0.03846153, 0.07211538],
      [0.02403846, 0.03125]]


def test_headline_oracle_job_runs_in_a_spawned_worker():
    from tf2_yolo_amd import graphs, labels
    hw, cls = 64, 3
    b = graphs.build_yolov3((hw, hw, 3), cls, anchors=A9)
    w = labels.synthetic_keras_weights(b, 7)
    x, ys = labels.synthetic_batch(np.random.default_rng(8), 2, (hw, hw), cls)
    ys = [np.asarray(y, dtype=np.float32) for y in ys]
    here = oracle_jobs.headline_job(3, hw // 32, A9, w, x, ys, 2)
    with ProcessPoolExecutor(max_workers=1, mp_context=mp.get_context("spawn"), initializer=oracle_jobs.worker_init) as pool:
        there = pool.submit(oracle_jobs.headline_job, 3, hw // 32, A9, w, x, ys, 2).result(timeout=300)
    assert [a.shape for a in here["ref"]] == [(2, 2, 2, 24), (2, 4, 4, 24), (2, 8, 8, 24)]
    assert all(np.isfinite(v) for v in here["ref_losses"] + here["l32"])
    for a, c in zip(here["ref"], there["ref"]):
        assert a.dtype == np.float64 and np.allclose(a, c, rtol=0, atol=1e-12)
    for a, c in zip(here["o32"], there["o32"]):
        assert a.dtype == np.float32 and np.allclose(a, c, rtol=1e-5, atol=1e-6)
    assert np.allclose(here["ref_losses"], there["ref_losses"], rtol=1e-12)
    assert set(here["moving"]) == set(there["moving"]) and len(here["moving"]) == 72
    # float32 against float64 on the same inputs: the "floor" the GPU tests scale their bars with
    floor = max(np.abs(a32.astype(np.float64) - a).max() / np.abs(a).max() for a, a32 in zip(here["ref"], here["o32"]))
    assert floor < 1e-3
