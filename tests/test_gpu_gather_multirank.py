"""qe_gather / qe_comm_* with MORE THAN ONE RANK, through the C ABI (BASELINE configs[4]'s exchange step).

RCCL refuses two ranks on one device and a GPU box of this pool has one GPU, so the ranks are fresh processes that share
GPU 0 and the bytes between them travel through tests/transport (a test-only stand-in for librccl selected with
QE_RCCL_LIBRARY: Unix sockets + host staging).  Everything else is the product: qe_filter_project on each rank's row-range
shard, the header all-gather, offsets, one grouped send / recv per column at the final offsets, bitmap words shifted into
place, the status exchange and the error paths of qe_comm.cpp -- none of which runs at world size 1.

Scenarios live in tests/gather_worker.py; each rank writes a report, the root compares against ONE pass over the whole
table on its own context, bit for bit."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRANSPORT = os.path.join(ROOT, "tests", "transport", "libqe_test_transport.so")

EXPECTED = ["comm_init", "allgather_host", "cfg2_nullable_ragged_empty_shard_root1", "cfg2_shard_range_root0",
            "overlapped_gather_ragged_root0", "overlapped_gather_root_last_16_slices", "overlapped_gather_empty_shard",
            "overlapped_gather_no_filter_mixed_validity", "mixed_validity",
            "shared_dictionary", "dictionary_mismatch_is_invalid_arg_on_every_rank", "schema_mismatch_is_invalid_arg_on_every_rank",
            "gather_after_errors", "root_out_of_range", "done"]


def run_ranks(world, timeout=420):
    if not os.path.exists(TRANSPORT):
        subprocess.run(["make", "-s", "-C", os.path.dirname(TRANSPORT)], check=True)
    tmp = tempfile.mkdtemp(prefix="qe_gather_")
    env = dict(os.environ, QE_RCCL_LIBRARY=TRANSPORT, QE_TEST_TRANSPORT_TIMEOUT_S="90", TMPDIR=tmp)
    idfile = os.path.join(tmp, "uid")
    procs = []
    for r in range(world):
        log = open(os.path.join(tmp, f"rank{r}.log"), "w")
        procs.append((subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "gather_worker.py"), str(r), str(world), idfile,
                                        os.path.join(tmp, f"report{r}.json")], env=env, stdout=log, stderr=subprocess.STDOUT), log))
    rcs = []
    try:
        for p, log in procs:
            rcs.append(p.wait(timeout=timeout))
            log.close()
    finally:
        for p, _ in procs:
            if p.poll() is None:
                p.kill()     # the exact processes this test started
    reports = []
    for r in range(world):
        path = os.path.join(tmp, f"report{r}.json")
        reports.append(json.load(open(path)) if os.path.exists(path) else [])
    logs = [open(os.path.join(tmp, f"rank{r}.log")).read()[-2000:] for r in range(world)]
    return rcs, reports, logs


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [3, 2])
def test_qe_gather_multirank_through_the_c_abi(native_lib, world):
    rcs, reports, logs = run_ranks(world)
    for r in range(world):
        names = [e["name"] for e in reports[r]]
        assert names == EXPECTED, f"rank {r}: {names}\n{logs[r]}"
        bad = [e for e in reports[r] if not e["ok"]]
        assert not bad, f"rank {r}: {bad}\n{logs[r]}"
    assert rcs == [0] * world, logs


@pytest.mark.timeout(900)
def test_bench_rehearsal_two_ranks_runs_the_exchange(native_lib):
    """bench.py's N > 1 code path (cfg 5's shape: rank r scans global rows [r * rows, (r + 1) * rows), max-over-ranks timing,
    then the materialising exchange timed apart) with two ranks on ONE GPU: torch.distributed over gloo for the harness,
    qe_gather through the C ABI over the test transport for the exchange.  The line must carry the gather report and the
    root must hold exactly the rows the two scans selected."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    if not os.path.exists(TRANSPORT):
        subprocess.run(["make", "-s", "-C", os.path.dirname(TRANSPORT)], check=True)
    env = dict(os.environ, QE_BENCH_REHEARSAL="1", QE_RCCL_LIBRARY=TRANSPORT, QE_TEST_TRANSPORT_TIMEOUT_S="90")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--rows", "40000000", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=800)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-1500:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["rows_total"] == 80_000_000 and out["scaling"] == "weak"
    g = out["gather"]
    assert "error" not in g, g
    assert g["rows_on_root"] == out["config"]["selected_rows_total"] and abs(g["rows_on_root"] / 80e6 - 0.05) < 0.001
    assert g["ms"] > 0 and g["scan_plus_gather_ms"] >= g["ms"] * 0.5
    ov = g["overlapped"]
    assert "error" not in ov, ov
    assert ov["rows_on_root"] == g["rows_on_root"] and ov["ms"] > 0
