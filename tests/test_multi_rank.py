"""N > 1 path of bench.py on CPU: two gloo ranks shard independent streams
(no data-path collective) and agree on the max-over-ranks time and the
whole-job throughput."""
import os
import socket
import sys

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
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = bench.shard_streams(8, world, rank)
    # per-rank "work": distinct streams, per-rank elapsed time; bench.py's own reduction: MAX of the times, AND of the
    # occupancy checks (rank 1 reports a failed check in the second call)
    dist.barrier()
    elapsed, ok = bench.rank_reduce(dist, "gloo", 0.5 + 0.25 * rank, True)
    _, ok2 = bench.rank_reduce(dist, "gloo", 0.1, rank == 0)
    assert ok and not ok2
    assert bench.rank_reduce(None, "gloo", 1.25, True) == (1.25, True)
    fps = bench.aggregate_fps(frames_per_rank=512, steps=4, world_size=world, elapsed_max_s=elapsed)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    q.put((rank, mine, elapsed, fps, gathered))
    dist.destroy_process_group()


def test_two_ranks_shard_streams_and_reduce_time():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, e0, f0, g0), (r1, s1, e1, f1, g1) = out
    assert s0 == [0, 2, 4, 6] and s1 == [1, 3, 5, 7]          # disjoint, complete
    assert sorted(g0[0] + g0[1]) == list(range(8))
    assert e0 == e1 == 0.75                                     # MAX over ranks
    assert f0 == f1 == 512 * 4 * 2 / 0.75                       # whole-job frames/s


def test_algorithmic_bytes_match_survey():
    sys.path.insert(0, ROOT)
    import bench
    per, total = bench.algorithmic_bytes(1920, 1080)
    assert total == 70649925
    assert per["k_bilateral"] == 2 * 6220800
    _, total4k = bench.algorithmic_bytes(3840, 2160)
    assert abs(total4k - 265479684) <= 16


def _run_bench(args, env_extra, timeout=180):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_refuses_a_rank_count_it_was_not_launched_with():
    """`--gpus N` must be the number of ranks: a launcher mismatch is an error, never a silent single-rank run that
    prints n_gpus: 1 (without a launcher bench.py starts the N ranks itself when the node has N GPUs)."""
    r = _run_bench(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in r.stderr and r.stdout.strip() == ""
    r = _run_bench(["--gpus", "4"], {})          # no launcher and no GPUs here: refuses before touching anything
    assert r.returncode != 0 and "--gpus 4 but only" in r.stderr and r.stdout.strip() == ""
    r = _run_bench(["--gpus", "0"], {})
    assert r.returncode != 0
