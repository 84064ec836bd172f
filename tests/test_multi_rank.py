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


class _FakePipe:
    """Stands in for BoardPipeline on a CPU-only host: `run` costs time in proportion to the frames it is given (rank 1
    is the slower GPU) and records what it was asked to do."""

    def __init__(self, stream_id, frames, us_per_frame):
        self.stream_id, self.frames, self.us = stream_id, frames, us_per_frame
        self.calls = []

    def run(self, slot0, count):
        import time
        assert 0 <= slot0 and count > 0 and slot0 + count <= self.frames
        self.calls.append((slot0, count))
        time.sleep(count * self.us * 1e-6)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # bench.py's own code path for K = 4 streams on N = 2 ranks, 512 frames in flight per GPU: the streams a rank owns,
    # the frames per stream, the step (runs of every stream interleaved), the timed region and the reduction
    K, F, steps, warmup = 4, 512, 3, 1
    mine, Fs = bench.plan_streams(K, world, rank, F)
    pipes = [_FakePipe(sid, Fs, 20 + 10 * rank) for sid in mine]
    step, splits, bounds = bench.make_step(pipes, Fs, chunk=64, splits=2)
    marks = []
    elapsed_local = bench.timed_steps(step, steps, warmup, dist.barrier, lambda: None, lambda: marks.append(sum(len(p.calls) for p in pipes)))
    elapsed, ok, frames_all = bench.rank_reduce(dist, "gloo", elapsed_local, True, Fs * len(pipes))
    _, ok2 = bench.rank_reduce(dist, "gloo", 0.1, rank == 0)  # rank 1 reports a failed occupancy check
    assert ok and not ok2
    assert bench.rank_reduce(None, "gloo", 1.25, True) == (1.25, True)
    assert bench.rank_reduce(None, "gloo", 1.25, True, 7) == (1.25, True, 7)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    q.put((rank, mine, Fs, splits, bounds, [p.calls for p in pipes], marks, elapsed_local, elapsed, frames_all, gathered))
    dist.destroy_process_group()


def test_two_ranks_shard_four_streams_through_bench_code():
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
    (r0, s0, fs0, sp0, b0, c0, m0, l0, e0, fr0, g0), (r1, s1, fs1, sp1, b1, c1, m1, l1, e1, fr1, g1) = out
    assert s0 == [0, 2] and s1 == [1, 3]                         # stream i on rank i % N: disjoint, complete
    assert sorted(g0[0] + g0[1]) == list(range(4))
    assert fs0 == fs1 == 256 and sp0 == 2 and b0 == [0, 128, 256]  # a GPU's 512 frames split evenly over its 2 streams
    # a step enqueues run k of EVERY stream before run k + 1 of any: (warm-up + 3 steps) x 2 runs per stream, interleaved
    for calls in c0 + c1:
        assert calls == [(0, 128), (128, 128)] * 4
    assert m0 == m1 == [4]                                        # the timed region starts after exactly the warm-up step
    assert l1 > l0 and e0 == e1 == max(l0, l1)                    # MAX over ranks of the EXACTLY-3-step region
    assert fr0 == fr1 == 1024                                     # SUM over ranks of the frames in flight
    assert 3 * 512 * (20 + 10) * 1e-6 <= e0 < 3 * 512 * 30e-6 * 3  # the slower rank's sleep time, not made up


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
