"""The multi-rank path of bench.py, rehearsed on ONE GPU (gloo, host-staged collectives).  Kept in its own file, collected
after test_gpu_parity.py: these tests start subprocesses (torch.distributed.run), and a rendezvous hiccup here must not hide the
kernel parity tests when the suite runs with -x."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ranks,exchange", [(2, "allgather"), (3, "neighbours"), (4, "neighbours")])
def test_multi_rank_bench_rehearsal(ranks, exchange):
    """bench.py's N > 1 path end to end with 2-4 ranks sharing this GPU (gloo, host-staged collectives): image sharding,
    exchange, block-pair scoring, all-to-all of candidate lists, merge -- each rank asserts that its lists equal the
    single-GPU ranking bit for bit.  Everything but the RCCL transport of the measured configuration."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from conftest import REPO
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PVS_BENCH_BACKEND="gloo", PVS_BENCH_EXCHANGE=exchange, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", str(ranks), "--steps", "1", "--warmup", "1",
           "--images", "1030", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stderr.count("identical to the single-GPU ranking") == ranks, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == ranks and "REHEARSAL" in line["backend"] and line["exchange"] == exchange


@pytest.mark.parametrize("retrieval", ["exact", "filtered"])
def test_one_rank_rccl_self_check(retrieval):
    """The SAME multi-rank code path on the measured transport: one rank, backend nccl (= RCCL) -- process group bound to the
    device, the engine on the stream RCCL synchronises with, the asynchronous all-gather overlapped with the (r, r) block,
    the all-to-all of the candidate lists -- checked bit for bit against the plain single-GPU retrieval."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from conftest import REPO
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PVS_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("PVS_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--images", "1030", "--no-cpu-baseline", "--retrieval", retrieval]
    r = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stderr.count("identical to the single-GPU ranking") == 1, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert "RCCL self-check" in line["backend"] and line["exchange_overlaps_own_block"] is True
