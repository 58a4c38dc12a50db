"""bench.py's partition of the job over the GPUs (no device needed).  BASELINE.json configs[4] is a FIXED total of
2 097 152 instances "sharded across 8 MI355X": the default is strong scaling, so `--gpus N` must total that for every N the
driver runs (1, 2, 4, 8) - the N = 1 point of the scaling curve is the configuration the metric is quoted on - and
`--scaling weak` keeps today's 262 144 per GPU.  Match: /root/reference/README.md:11-13 (the multi-instance claim)."""
import json
import os
import subprocess
import sys

import pytest

import bench
import fx8010_programs as progs
import fx8010_shard as shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_strong_scaling_totals_the_stated_configuration(world):
    spans = [bench.plan_instances("config5", "strong", 0, world, r, progs, shard) for r in range(world)]
    assert all(t == 2097152 for _, _, t in spans)
    assert sum(c for _, c, _ in spans) == 2097152
    assert spans[0][0] == 0
    for (f0, c0, _), (f1, _, _) in zip(spans[:-1], spans[1:]):
        assert f0 + c0 == f1                       # contiguous ranges: the stimulus of instance n does not depend on N
    assert {c for _, c, _ in spans} == {2097152 // world}
    if world == 8:                                  # ... and the 8-GPU job is the same under either scaling
        assert spans == [bench.plan_instances("config5", "weak", 0, 8, r, progs, shard) for r in range(8)]


def test_ragged_totals_and_the_other_configurations():
    for total in (2097152 - 37, 1000003, 65):
        for world in (1, 2, 3, 4, 8):
            spans = [bench.plan_instances("config5", "strong", total, world, r, progs, shard) for r in range(world)]
            assert sum(c for _, c, _ in spans) == total and max(c for _, c, _ in spans) - min(c for _, c, _ in spans) <= 1
            assert all(spans[i][0] + spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    # the single-GPU configurations: their BASELINE count is the whole job
    for cfg, n in (("config2", 4096), ("config3", 65536), ("config4", 262144)):
        assert bench.plan_instances(cfg, "strong", 0, 1, 0, progs, shard) == (0, n, n)
    # weak: the per-GPU share on every rank, offsets so that no two ranks emulate the same instance
    assert bench.plan_instances("config5", "weak", 0, 4, 3, progs, shard) == (3 * 262144, 262144, 4 * 262144)
    assert bench.plan_instances("config5", "weak", 1000, 2, 1, progs, shard) == (1000, 1000, 2000)
    with pytest.raises(SystemExit):
        bench.plan_instances("config5", "strong", 3, 8, 0, progs, shard)


def test_the_default_is_strong_scaling_over_configs4():
    saved = sys.argv
    try:
        sys.argv = ["bench.py"]
        a = bench.parse()
    finally:
        sys.argv = saved
    assert a.scaling == "strong" and a.config == "config5" and a.gpus == 1 and a.instances == 0
    assert progs.CONFIG_TOTAL_INSTANCES["config5"] == 2097152 and progs.CONFIG_INSTANCES["config5"] * 8 == 2097152
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "2097152 instances sharded across 8" in base["configs"][4]
