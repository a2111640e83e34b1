import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # the two full-batch headline tests go last: their CPU oracle passes run in worker processes meanwhile (below)
    sel = [it for it in items if "test_headline_configs_at_their_true_batch_vs_fp64_oracle" in it.nodeid]
    if sel:
        items[:] = [it for it in items if it not in sel] + sel
    # `-m gpu` on a box without a GPU should fail loudly, not silently skip: only skip gpu tests
    # when they were NOT explicitly selected.
    if _has_gpu():
        return
    mexpr = config.getoption("-m") or ""
    if "gpu" in mexpr and "not gpu" not in mexpr:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ---- the full-batch headline tests (tests/test_gpu_fullsize.py::test_headline_configs_at_their_true_batch_vs_fp64_oracle) ----
# Their float64 / float32 CPU oracle passes (C4: ~2 minutes of host time, C3: ~1) are independent of anything the device
# computes, so a GPU session starts them in two worker processes right away (tests/oracle_jobs.py) and runs the two tests
# LAST: the host computes the oracle while the device runs the rest of the suite.
HEADLINE_JOBS = {}      # config -> (future, digest of the job's inputs)


def foreground_threads():
    """torch's CPU threads for the oracle passes the tests run in THIS process, set explicitly: the box's cores (at most 16)
    once the background jobs are done, half of them while they run. Called by the tests that spend their time on the host."""
    import torch
    ncpu = min(16, os.cpu_count() or 8)
    busy = any(not f.done() for f, _ in HEADLINE_JOBS.values())
    n = max(4, ncpu // 2) if busy else ncpu
    if torch.get_num_threads() != n:
        torch.set_num_threads(n)
    return n


def _headline_selected(items):
    return [it for it in items if "test_headline_configs_at_their_true_batch_vs_fp64_oracle" in it.nodeid]


@pytest.hookimpl(tryfirst=True)
def pytest_runtestloop(session):
    """starts the two oracle jobs when the test loop begins -- not at collection time (`--collect-only` builds nothing and
    starts nothing: ADVICE r05) -- and returns None, so that pytest's own loop runs the tests"""
    if session.config.option.collectonly or session.testsfailed:
        return None
    _start_headline_jobs(session)
    return None


def _start_headline_jobs(session):
    items = session.items
    sel = _headline_selected(items)
    if not sel or not _has_gpu():
        return
    import torch
    ncpu = os.cpu_count() or 8
    foreground_threads()
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    import oracle_jobs
    import test_gpu_model as T
    pool = ProcessPoolExecutor(max_workers=2, mp_context=mp.get_context("spawn"), initializer=oracle_jobs.worker_init)
    session._headline_pool = pool
    want = {it.callspec.params["config"] for it in sel}
    for config in ("C4", "C3"):                # the longer job first
        if config not in want:
            continue
        version, hw, N = (3, 416, 32) if config == "C3" else (4, 608, 16)
        y, model, fwd, loss_o, loss_g, x, ys = T._setup(version, hw=hw, N=N, class_num=T.HEADLINE_CLASSES)
        w = T._weights_dict(model)
        threads = max(2, min(16, ncpu) // 4)
        HEADLINE_JOBS[config] = (pool.submit(oracle_jobs.headline_job, version, hw // 32, T.A9, w, x, ys, threads,
                                             T.HEADLINE_CLASSES),
                                 T.inputs_digest(w, x, ys))
        del y, model, w
        torch.cuda.empty_cache()


def pytest_sessionfinish(session, exitstatus):
    pool = getattr(session, "_headline_pool", None)
    if pool is not None:
        pool.shutdown(wait=False, cancel_futures=True)
