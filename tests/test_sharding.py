"""The N > 1 path on CPU: world_size-2 gloo processes exercise the stream sharding and the
counter reduction that bench.py runs over RCCL."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import nunet_amd  # noqa: F401
    from nunet_amd.sharding import collective_proof, reduce_throughput, stream_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = stream_range(2048 + 3, rank, world)
    frames, elapsed = (hi - lo) * 10, 1.0 + rank          # rank 1 is the slow one
    tot, mx = reduce_throughput(frames, elapsed, dist, torch.device("cpu"))
    proof = collective_proof(frames / elapsed, dist, torch.device("cpu"))
    assert proof["ranks_reduced"] == world and proof["backend"] == "gloo"
    want = [(stream_range(2048 + 3, r, world)[1] - stream_range(2048 + 3, r, world)[0]) * 10 / (1.0 + r) for r in range(world)]
    assert proof["per_rank"] == want, (proof["per_rank"], want)      # (the same expression every rank evaluated: exact)
    q.put((rank, lo, hi, tot, mx))
    dist.destroy_process_group()


def test_stream_range_partitions_exactly():
    from nunet_amd.sharding import stream_range
    for total, world in ((2048, 8), (10, 3), (1, 2), (0, 4)):
        spans = [stream_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        stream_range(8, 2, 2)


@pytest.mark.timeout(120)
def test_counter_reduction_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    (r0, lo0, hi0, tot0, mx0), (r1, lo1, hi1, tot1, mx1) = res
    assert (lo0, hi1) == (0, 2051) and hi0 == lo1
    assert tot0 == tot1 == 2051 * 10          # whole-job frames
    assert mx0 == mx1 == 2.0                  # max over ranks


@pytest.mark.timeout(300)
@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_py_rank_plumbing_two_ranks_gloo(launcher):
    """`python bench.py --gpus 2` with no launcher around it must spawn its two ranks itself (and keep working under
    torch.distributed.run, which is how the driver starts it): self-launch, rank -> stream slice, barrier, counter
    reduction and the single JSON line of rank 0 -- on CPU over gloo with a stand-in engine (no model arithmetic)."""
    import json
    import subprocess
    args = ["--gpus", "2", "--steps", "5", "--warmup", "2", "--batch", "6", "--selftest-launcher", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    if launcher == "self":
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout            # ONE line, from rank 0
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 5 and j["warmup"] == 2 and j["scaling"] == "weak"
    assert j["config"]["streams_per_gpu"] == 6 and j["config"]["total_streams"] == 12
    assert j["unit"] == "frames/s" and j["value"] > 0
    # whole-job frames / max-over-ranks time (ms_per_step is printed with 4 decimals: allow for its rounding)
    assert abs(j["value"] - 12 * 5 / (j["ms_per_step"] * 5e-3)) / j["value"] < 1e-3 + 6e-5 / j["ms_per_step"]
    # the line proves its own collective: an all-reduce of ones counted both ranks, over the backend it names, and the
    # all-gathered per-rank rates are two different measurements
    col = j["collective"]
    assert col["ranks_reduced"] == 2 and col["backend"] == "gloo"
    assert len(col["per_rank_frames_per_s"]) == 2 and all(r > 0 for r in col["per_rank_frames_per_s"])
    assert col["per_rank_frames_per_s"][0] != col["per_rank_frames_per_s"][1]
    assert j["short_window"] is True and j["timed_window_ms"] < 100.0          # 5 steps of the stand-in engine


@pytest.mark.timeout(300)
def test_counter_reduction_world8_gloo_ragged_totals():
    """The width the driver will run at (`--gpus 8`): eight gloo ranks, a stream total that does not divide by eight
    (2051 = 8 * 256 + 3: ranks 0..2 own 257 streams, the rest 256), SUM / MAX reductions and the all-gathered proof."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == list(range(8))
    assert res[0][1] == 0 and res[-1][2] == 2051 and all(res[i][2] == res[i + 1][1] for i in range(7))
    assert [r[2] - r[1] for r in res] == [257, 257, 257, 256, 256, 256, 256, 256]
    assert all(r[3] == 2051 * 10 for r in res)          # whole-job frames on every rank
    assert all(r[4] == 8.0 for r in res)                # max over ranks: rank 7 took 1 + 7 s


@pytest.mark.timeout(300)
def test_bench_py_rank_plumbing_eight_ranks_gloo():
    """bench.py exactly as the driver starts it for the 8-GPU scaling point (torch.distributed.run, 8 ranks, 127.0.0.1), on CPU
    over gloo with the stand-in engine: one JSON line, `collective.ranks_reduced == 8`, eight per-rank rates, weak scaling."""
    import json
    import subprocess
    args = ["--gpus", "8", "--steps", "4", "--warmup", "1", "--batch", "5", "--selftest-launcher", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["scaling"] == "weak"
    assert j["config"]["streams_per_gpu"] == 5 and j["config"]["total_streams"] == 40
    col = j["collective"]
    assert col["ranks_reduced"] == 8 and col["backend"] == "gloo" and len(col["per_rank_frames_per_s"]) == 8
    assert "other_configs" not in j and "cpu_baseline" not in j          # N > 1: only the sharded workload is run


def test_bench_py_defines_every_entry_its_main_dispatches_to():
    """`bench.py --offline` / `other_configs` / `cpu_baseline` are separate functions that only a GPU run reaches: at least their names must
    resolve (an edit once dropped `bench_offline` and only the GPU box would have noticed)."""
    sys.path.insert(0, ROOT)
    import bench
    for fn in ("bench_offline", "other_config_records", "cpu_baseline", "kernel_report", "parity_check", "self_launch"):
        assert callable(getattr(bench, fn)), fn
