"""The N>1 path: trials sharded over ranks by global index, one gather at the end.
world_size-2 `gloo` run on CPU (decoders injected from the oracle) must reproduce the
single-process result trial for trial."""
import importlib
import os
import socket

import numpy as np
import pytest

from helpers import S

shard = importlib.import_module("sca-ldpc_amd.shard")
trials = importlib.import_module("sca-ldpc_amd.trials")


def test_trial_range_partitions():
    for total in (0, 1, 7, 64, 1000):
        for world in (1, 2, 3, 8):
            spans = [shard.trial_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _instance():
    N, W, R, omega, eps = 211, 5, 90, 3, 0.02
    rng = np.random.RandomState(3)
    sup = S.codes.make_random_ldpc_first_row(N, W, rng)
    Hin = S.codes.hqc_check_graph(sup, N, rng.permutation(N)[:R])
    return N, omega, eps, Hin


def _decode_range(start, stop):
    from oracle import pyoracle

    N, omega, eps, Hin = _instance()
    msg, ys = trials.hqc_trials(Hin, omega, eps, stop - start, base_seed=2, first_index=start)
    r = pyoracle.bp_decode_batch(Hin.with_identity(), trials.hqc_priors(N, Hin.m, omega, eps), msg, 1, 20, "min_sum",
                                 dtype="f32", threads=2)
    return trials.success(r["bits"], ys, N).astype(np.uint8)


def _worker(rank, world, port, total, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = shard.trial_range(total, rank, world)
    full = shard.gather_results(_decode_range(a, b), total, rank, world)
    if rank == 0:
        q.put(full)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process():
    import torch.multiprocessing as mp

    total = 37  # ragged: 19 + 18
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    full = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    single = _decode_range(0, total)
    assert np.array_equal(full, single)
    assert 0 < single.sum() < total  # both outcomes present


@pytest.mark.parametrize("world", [2, 3])
def test_success_checksum_is_independent_of_the_world_size(world):
    """BASELINE config 5's line carries `success_checksum` (which trials failed): identical for 1, 2 and 3 ranks,
    ragged splits included (41 = 21 + 20 = 14 + 14 + 13)."""
    import torch.multiprocessing as mp

    total = 41
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    full = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    single = _decode_range(0, total)
    assert shard.success_checksum(full) == shard.success_checksum(single) and shard.success_checksum(single) > 0
    assert np.array_equal(full, single)


def _run_bench(*argv, **env):
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, **env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)  # a plain shell, as the driver's `python3 bench.py --gpus N`
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=e, capture_output=True, text=True,
                          timeout=600)


def test_bench_launches_its_own_ranks_from_a_plain_shell():
    """`python bench.py --gpus 2` without torchrun around it must start the two ranks itself (as a child
    process, before anything touches a GPU), rendezvous on 127.0.0.1 and run the end-of-run collective;
    rank 0's line reports how many ranks the all_gather saw."""
    import json

    r = _run_bench("--gpus", "2", "--rendezvous-only", BENCH_BACKEND="gloo")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["backend"] == "gloo"


def test_bench_falls_back_to_gloo_when_rccl_does_not_come_up():
    """The first real multi-GPU run must not die on the transport: gloo comes up first, RCCL is tried
    for the gathers, and if it fails on any rank -- forced here, as there is no GPU -- every rank gathers over gloo in
    the same process and the line says so.  Exit code 0."""
    import json

    r = _run_bench("--gpus", "2", "--rendezvous-only", SCALDPC_FORCE_NCCL_FAILURE="1")  # BENCH_BACKEND defaults to nccl
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["backend"] == "gloo"
    assert "nccl failed, gathers over gloo" in line["collective_note"]


def _collectives_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SCALDPC_FORCE_NCCL_FAILURE"] = "1"
    coll = shard.Collectives(rank, world, device=None, want="nccl")
    got = coll.gather(np.arange(3 if rank == 0 else 2, dtype=np.int32) + 10 * rank, 5)  # ragged: 3 + 2
    times = coll.gather_scalars(1.5 + rank)
    coll.barrier()
    if rank == 0:
        q.put((coll.backend, coll.note, got.tolist(), times))
    coll.close()


def test_collectives_fall_back_inside_the_process():
    """shard.Collectives on two CPU ranks: the RCCL attempt fails on both (no GPU), the ranks agree over gloo and the
    gathers -- ragged result vectors, per-rank scalars -- run there."""
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_collectives_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    backend, note, got, times = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert backend == "gloo" and "nccl failed" in note and got == [0, 1, 2, 10, 11] and times == [1.5, 2.5]


def test_bench_forwards_its_childrens_failure():
    """No GPU here: the ranks refuse to run (no CPU fallback in the product path) and the launcher's
    exit code says so -- the self-launch must not swallow a failed rank."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    r = _run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", BENCH_BACKEND="gloo")
    assert r.returncode != 0
    assert "needs a GPU" in (r.stderr + r.stdout)
