"""Multi-process path on CPU: two gloo ranks shard the instances exactly as bench.py shards them
over GPUs (contiguous ranges, counter-based stimulus offset by the shard's first instance, no
data-path collective) and must reproduce the single-process result; the timing/counter
reductions bench.py relies on are exercised too.  The per-shard compute here is the CPU oracle
standing in for the GPU kernel: what is under test is the sharding and the reductions."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fx8010_programs as progs
import fx8010_shard as shard
from pyoracle import Oracle


def test_shard_ranges_cover_everything():
    for n in (1, 2, 7, 64, 1000, 2097152):
        for w in (1, 2, 3, 4, 8):
            spans = [shard.shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (f0, c0), (f1, _) in zip(spans[:-1], spans[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    assert shard.weak_shard(262144, 3) == (786432, 262144)


def test_library_partition_for_up_to_eight_devices():
    """The C++ host's partition (fx_shard.cpp Sharded::plan, through the C ABI - no device needed): contiguous, complete, whole
    wavefronts on every shard but the last, balanced to one wavefront, for 1..8 devices and ragged instance counts; and the
    sizes BASELINE.json names."""
    import fx8010_amd as A
    sizes = [1, 2, 63, 64, 65, 127, 128, 129, 511, 512, 513, 1000, 4096, 65536, 65553, 262144, 262145, 2097152, 2097152 + 17, 3 * 2097152 + 1]
    for n in sizes:
        for k in range(1, 9):
            plan = A.shard_plan(n, k)
            waves = (n + 63) // 64
            if plan is None:
                assert waves < k or n < k, (n, k)     # refused only when a shard would be empty
                continue
            assert len(plan) == k and plan[0][0] == 0 and sum(c for _, c in plan) == n, (n, k)
            for (f0, c0), (f1, c1) in zip(plan[:-1], plan[1:]):
                assert f0 + c0 == f1 and c0 % 64 == 0 and f1 % 64 == 0, (n, k, plan)   # shards start on a wavefront
            assert all(c >= 1 for _, c in plan)
            per = [(c + 63) // 64 for _, c in plan]
            assert max(per) - min(per) <= 1, (n, k, plan)                             # balanced to one wavefront
    assert A.shard_plan(2097152, 8) == [(i * 262144, 262144) for i in range(8)]        # BASELINE configs[4]
    assert A.shard_plan(64, 2) is None and A.shard_plan(0, 1) is None and A.shard_plan(10, 0) is None


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_total, n_samples, text, result_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    d = shard.init_process_group("gloo")
    assert d is not None and d.get_world_size() == world
    first, count = shard.shard_range(n_total, world, rank)
    x = progs.stimulus(count, n_samples, first_instance=first)
    y = np.empty_like(x)
    executed = 0
    for i in range(count):
        o = Oracle(1)
        assert o.load_text(text)
        y[:, i] = o.process_block(x[:, i].copy())
        executed += o.instruction_counter()
    d.barrier()
    elapsed = shard.reduce_scalar(d, 1.0 + rank, "max")          # bench.py: MAX over ranks of the timed region
    total = shard.reduce_scalar(d, executed, "sum")               # bench.py: executed instructions of the whole job
    per_rank = shard.gather_scalars(d, 10.0 + rank)               # bench.py: every device's own kernel time (per_gpu_kernel_ms)
    assert per_rank == [10.0 + r for r in range(world)]
    gathered = [None] * world
    d.all_gather_object(gathered, (first, y))
    if rank == 0:
        full = np.concatenate([g[1] for g in sorted(gathered, key=lambda g: g[0])], axis=1)
        np.save(result_path, full)
        with open(result_path + ".txt", "w") as fh:
            fh.write("%r %r" % (elapsed, total))
    d.barrier()
    d.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_two_ranks_reproduce_one(tmp_path, world):
    n_total, n_samples = 10, 48  # uneven split exercised with world=2? 10 -> 5+5; use 9 below as well
    text = progs.config4()
    path = str(tmp_path / "out.npy")
    mp.spawn(_worker, args=(world, _free_port(), n_total, n_samples, text, path), nprocs=world, join=True)
    got = np.load(path)
    x = progs.stimulus(n_total, n_samples)
    executed = 0
    for i in range(n_total):
        o = Oracle(1)
        assert o.load_text(text)
        assert np.array_equal(o.process_block(x[:, i].copy()).view(np.uint32), got[:, i].view(np.uint32)), i
        executed += o.instruction_counter()
    elapsed, total = [float(v) for v in open(path + ".txt").read().split()]
    assert elapsed == 2.0 and total == float(executed)
    assert shard.gather_scalars(None, 3.5) == [3.5]   # a single process: its own time
