"""Host file work of a tile next to the device work of the next one (round 6).

A 1 M-point tile is ~60 ms of device work between ~1.2 s of host files in the reference's order of work: two PLY reads, two
partition text files (`x y z r g b label`, xyz_io.h:192-221), four to eight result tables (`save_process_dvf`,
src/coarse_to_fine_matching_base.py:3459-3600).  None of the files is read back by the same run (the labels stay on the device,
`load_partition`), so the entry points hand them to a small pool of writer threads -- the C writers release the GIL -- and read the
NEXT tile's PLY files ahead on the same pool.  What lands on disk is byte for byte what the serial order of work writes; the run
waits for the writers before it returns (`drain`), and a writer's exception is raised there (or at the next `submit`).

F4L_ASYNC_IO=0 switches the pool off: every job runs where it is submitted (the reference's order of work, for A/B and debugging).
"""
import os
import threading
from concurrent.futures import ThreadPoolExecutor

_LOCK = threading.Lock()
_POOL = None
_PENDING = []
_PREFETCH = {}


def enabled():
    return os.environ.get("F4L_ASYNC_IO", "1") != "0"


def _pool():
    global _POOL
    with _LOCK:
        if _POOL is None:
            _POOL = ThreadPoolExecutor(max_workers=max(2, min(4, (os.cpu_count() or 2) // 2)), thread_name_prefix="f4l-io")
        return _POOL


class _Done:
    """A finished job (the synchronous mode): the interface of a Future, nothing pending."""

    def __init__(self, value):
        self._value = value

    def result(self):
        return self._value


def _raise_finished():
    """Re-raises the exception of a writer that has failed since the last look, and forgets the jobs that are through."""
    with _LOCK:
        done = [f for f in _PENDING if f.done()]
        _PENDING[:] = [f for f in _PENDING if not f.done()]
    for f in done:
        f.result()


def submit(fn, *args, **kw):
    """Runs `fn(*args, **kw)` on a writer thread (its arguments must stay untouched until it is through: hand over arrays nobody
    writes to any more).  Returns a future; `drain()` waits for all of them."""
    if not enabled():
        return _Done(fn(*args, **kw))
    _raise_finished()
    fut = _pool().submit(fn, *args, **kw)
    with _LOCK:
        _PENDING.append(fut)
    return fut


def drain():
    """Waits for every submitted job; raises the first exception any of them ended with."""
    with _LOCK:
        jobs = list(_PENDING)
        _PENDING.clear()
    first = None
    for f in jobs:
        try:
            f.result()
        except BaseException as e:  # noqa: BLE001  (every job is waited for before the first failure is raised)
            first = first or e
    if first is not None:
        raise first


def prefetch(path, reader):
    """Starts `reader(path)` ahead of its use; `take(path, reader)` returns its result (or reads now when nothing was started)."""
    if not enabled() or path in _PREFETCH:
        return
    fut = _pool().submit(reader, path)
    with _LOCK:
        _PREFETCH[path] = fut


def take(path, reader):
    with _LOCK:
        fut = _PREFETCH.pop(path, None)
    return reader(path) if fut is None else fut.result()


def forget_prefetched():
    """Drops reads that were started and never taken (a run that ended early)."""
    with _LOCK:
        futs = list(_PREFETCH.values())
        _PREFETCH.clear()
    for f in futs:
        try:
            f.result()
        except BaseException:  # noqa: BLE001
            pass
