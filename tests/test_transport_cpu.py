"""Self-test of tests/transport/ (the test-only stand-in for librccl that lets qe_gather run with several ranks on one
GPU): its exchange engine on host memory, three processes, before any GPU test relies on it.  All-gather in rank order;
a grouped gather-to-root with several operations per pair, payloads far larger than a socket buffer, a zero-byte and an
absent contribution; a peer that never posts is an error after the timeout, not a hang."""
import ctypes as C
import multiprocessing as mp
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRANSPORT = os.path.join(ROOT, "tests", "transport", "libqe_test_transport.so")


def _build():
    subprocess.run(["make", "-s", "-C", os.path.dirname(TRANSPORT)], check=True)


def _payload(rank, k, n):
    return (np.arange(n, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(rank * 1000 + k)).view(np.uint8)


SIZES = [0, 8, 3 << 20, 12345]     # bytes per operation; the 3 MiB one exceeds any Unix-socket buffer


def _worker(rank, world, uid, q, stall_rank):
    os.environ["QE_TEST_TRANSPORT_HOSTMEM"] = "1"
    os.environ["QE_TEST_TRANSPORT_TIMEOUT_S"] = "3" if stall_rank is not None else "30"
    L = C.CDLL(TRANSPORT)

    class Uid(C.Structure):
        _fields_ = [("b", C.c_char * 128)]
    L.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
    for f in (L.ncclSend, L.ncclRecv):
        f.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    L.ncclGetErrorString.restype = C.c_char_p
    comm = C.c_void_p()
    u = Uid()
    u.b = uid
    try:
        assert L.ncclCommInitRank(C.byref(comm), world, u, rank) == 0
        mine = np.full(40, rank + 1, dtype=np.uint8)
        allg = np.zeros(40 * world, dtype=np.uint8)
        assert L.ncclAllGather(mine.ctypes.data, allg.ctypes.data, 40, 1, comm, None) == 0
        ok = all((allg[40 * r:40 * (r + 1)] == r + 1).all() for r in range(world))
        root = 1
        st = 0
        if stall_rank is not None and rank == stall_rank:
            q.put((rank, "stalled", True))         # posts nothing: the others must time out with an error
        elif rank == root:
            bufs = {}
            assert L.ncclGroupStart() == 0
            for p in range(world):
                if p == root:
                    continue
                for k, n in enumerate(SIZES):
                    if p == 2 and k == 3:
                        continue                   # rank 2 contributes one operation fewer
                    bufs[p, k] = np.zeros(max(n, 1), dtype=np.uint8)
                    assert L.ncclRecv(bufs[p, k].ctypes.data, n, 1, p, comm, None) == 0
            st = L.ncclGroupEnd()
            if stall_rank is None:
                assert st == 0, L.ncclGetErrorString(st)
                for (p, k), b in bufs.items():
                    n = SIZES[k]
                    ok = ok and np.array_equal(b[:n], _payload(p, k, (n + 7) // 8)[:n])
        else:
            keep = []
            assert L.ncclGroupStart() == 0
            for k, n in enumerate(SIZES):
                if rank == 2 and k == 3:
                    continue
                src = np.ascontiguousarray(_payload(rank, k, (n + 7) // 8)[:max(n, 1)]) if n else np.zeros(1, np.uint8)
                keep.append(src)
                assert L.ncclSend(src.ctypes.data, n, 1, root, comm, None) == 0
            st = L.ncclGroupEnd()
            if stall_rank is None:
                assert st == 0, L.ncclGetErrorString(st)
        if stall_rank is None or rank != stall_rank:
            q.put((rank, st, bool(ok)))
        if stall_rank is not None and rank == stall_rank:
            import time
            time.sleep(5)                          # keep the sockets open while the others wait
        L.ncclCommDestroy(comm)
    except Exception as exc:   # noqa: BLE001
        q.put((rank, f"{type(exc).__name__}: {exc}", False))


def _run(stall_rank=None):
    _build()
    L = C.CDLL(TRANSPORT)
    uid = C.create_string_buffer(128)
    assert L.ncclGetUniqueId(uid) == 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 3
    ps = [ctx.Process(target=_worker, args=(r, world, uid.raw.split(b"\0")[0], q, stall_rank)) for r in range(world)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=60) for _ in range(world))
    for p in ps:
        p.join(20)
        assert not p.is_alive()
    return out


@pytest.mark.timeout(120)
def test_transport_allgather_and_grouped_gather_to_root():
    assert _run() == [(0, 0, True), (1, 0, True), (2, 0, True)]


@pytest.mark.timeout(120)
def test_transport_missing_peer_is_an_error_not_a_hang():
    out = dict((r, (st, ok)) for r, st, ok in _run(stall_rank=0))
    assert out[0] == ("stalled", True)
    assert out[1][0] != 0          # the root waited for rank 0's sends: system error after the timeout
    assert out[2] == (0, True)     # rank 2's sends were taken by the root
