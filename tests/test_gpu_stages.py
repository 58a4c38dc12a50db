"""Programs pipelined over the wavefronts of a workgroup (fx_xlate.hpp StageInfo) on the device: bit for bit against the oracle
whatever the number of stages - outputs, registers, instruction counters, state carried over block boundaries, blocks shorter
than the pipeline, non-finite values crossing a cut."""
import os
import sys

import numpy as np
import pytest

import fx8010_programs as progs
from pyoracle import Oracle

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.fixture
def stages(monkeypatch):
    def pin(k):
        monkeypatch.delenv("FX_KERNEL", raising=False)
        monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
        monkeypatch.delenv("FX_BUILDER", raising=False)        # (tests that want the caller's thread only say so themselves)
        monkeypatch.delenv("FX_STAGES_TUNE", raising=False)
        if k is None:
            monkeypatch.delenv("FX_STAGES", raising=False)
        else:
            monkeypatch.setenv("FX_STAGES", str(k))
    return pin


def run_and_compare(gpu, text, x, blocks, regs, pre=None):
    N = x.shape[1]
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(text), b.errors()
    if pre:
        pre(b, None)
    ys = [b.process_block(x[lo:hi]) for lo, hi in blocks]
    for n in range(N):
        o = Oracle(1)
        assert o.load_text(text)
        if pre:
            pre(None, (o, n))
        for (lo, hi), y in zip(blocks, ys):
            r = o.process_block(x[lo:hi, n].copy())
            assert np.array_equal(r.view(np.uint32), np.ascontiguousarray(y[:, n]).view(np.uint32)), "instance %d block %d:%d" % (n, lo, hi)
        assert b.instruction_counter_i(n) == o.instruction_counter(), n
        for reg in regs:
            assert b.get_register_bits_i(reg, n) == o.get_register_bits(reg), (reg, n)
    assert b.ood_flags() == 0
    return b


@pytest.mark.parametrize("k", [2, 3, 4, 8, 16])
def test_config2_in_stages(gpu, stages, k):
    stages(k)
    N, S = 200, 83   # four workgroups, the last one ragged (8 instances); blocks of 83, 1, 2, 5 and 30 samples
    x = progs.stimulus(N, S + 38)
    blocks = [(0, S), (S, S + 1), (S + 1, S + 3), (S + 3, S + 8), (S + 8, S + 38)]
    b = run_and_compare(gpu, progs.config2(), x, blocks, ["t", "s0", "s7", "s15", "s30", "out", "ccr", "in"])
    assert b.info("waves_per_wg") == k and b.info("kernel") >= 9


@pytest.mark.parametrize("group", [1, 2, 4])
def test_rings_shorter_than_the_pipeline(gpu, stages, group, monkeypatch):
    """the packet ring has 4 x group buffers; with little LDS per workgroup (large batches, wide packets) group drops to 2 or 1 and
    the ring is shorter than the pipeline has stages: the first request of stage k >= ring must wrap like every later one
    (found at 98 304 instances x 8 stages: stages 4 .. 7 read their first packet from beyond the ring)"""
    stages(8)
    monkeypatch.setenv("FX_STAGES_GROUP", str(group))
    N, S = 200, 50
    x = progs.stimulus(N, S + 9)
    b = run_and_compare(gpu, progs.config2(), x, [(0, S), (S, S + 1), (S + 1, S + 9)], ["t", "s0", "s7", "s15", "s30", "out", "ccr", "in"])
    assert b.info("waves_per_wg") == 8


@pytest.mark.parametrize("k,group", [(8, None), (8, 1), (3, 2), (16, None)])
def test_every_block_length(gpu, stages, k, group, monkeypatch):
    """blocks of 1, 2, 3 .. 18 samples one after the other (state carried): every position of a block's end inside a group of
    samples, blocks shorter than a group, shorter than the pipeline is deep"""
    stages(k)
    if group:
        monkeypatch.setenv("FX_STAGES_GROUP", str(group))
    N = 70
    cuts = [0]
    for n in range(1, 19):
        cuts.append(cuts[-1] + n)
    x = progs.stimulus(N, cuts[-1])
    run_and_compare(gpu, progs.config2(), x, list(zip(cuts[:-1], cuts[1:])), ["t", "s0", "s9", "s30", "out", "ccr"])


def wide_packet_program(W):
    return progs.wide_chains(W)


@pytest.mark.parametrize("width", [4, 12])
def test_wide_packets(gpu, stages, width):
    """`width` parallel one-pole chains advance side by side: 5 or 13 rows cross every cut.  With 13 rows a buffer takes 32 KB and
    the LDS holds a ring of four for eight stages (group = 1) - by the default policy, no knob"""
    stages(None)
    W = width
    text = wide_packet_program(W)
    N, S = 130, 345
    x = progs.stimulus(N, S)
    b = run_and_compare(gpu, text, x, [(0, 300), (300, 301), (301, S)], ["a0", "a%d" % (W - 1), "s0_0", "s%d_5" % (W - 1), "out", "ccr"])
    assert b.info("waves_per_wg") == 8


def compare_sampled(b, text, x, cuts, ys, picks, regs):
    from concurrent.futures import ThreadPoolExecutor

    def one(n):
        o = Oracle(1)
        assert o.load_text(text)
        for (lo, hi), y in zip(zip(cuts[:-1], cuts[1:]), ys):
            r = o.process_block(x[lo:hi, n].copy())
            if not np.array_equal(r.view(np.uint32), np.ascontiguousarray(y[:, n]).view(np.uint32)):
                return "instance %d block %d:%d differs" % (n, lo, hi)
        return o.instruction_counter(), [o.get_register_bits(r) for r in regs]

    with ThreadPoolExecutor(max_workers=8) as pool:   # (the oracle releases the GIL)
        res = list(pool.map(one, picks))
    for n, r in zip(picks, res):
        assert not isinstance(r, str), r
    for n, r in list(zip(picks, res))[:: max(1, len(picks) // 48)]:   # (each read is a round trip to the device)
        assert b.instruction_counter_i(n) == r[0], n
        assert [b.get_register_bits_i(reg, n) for reg in regs] == r[1], n


def test_config2_at_its_baseline_shape(gpu, stages):
    """BASELINE configs[1] as bench.py times it: 4 096 instances of the 64-instruction chain in blocks of >= 300 samples, the
    library's own choice of stages (eight wavefronts per workgroup, a barrier every eighth sample, the long-block class) -
    EVERY instance against the oracle over three blocks, state carried"""
    stages(None)
    text = progs.config2()
    N, cuts = 4096, [0, 300, 301, 777]
    x = progs.stimulus(N, cuts[-1])
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(text), b.errors()
    ys = [b.process_block(x[lo:hi]) for lo, hi in zip(cuts[:-1], cuts[1:])]
    assert b.info("waves_per_wg") == 8 and b.info("kernel") >= 9 and b.info("xlate_builds") == 1
    compare_sampled(b, text, x, cuts, ys, list(range(N)), ["t", "s0", "s15", "s30", "out", "ccr"])
    assert b.ood_flags() == 0
    assert b.instruction_counter() == N * 64 * cuts[-1]


@pytest.mark.parametrize("program", ["config2", "wide12"])
@pytest.mark.parametrize("N", [4096, 32768, 98304, 114688])
def test_default_policy_across_batch_sizes(gpu, stages, program, N):
    """no FX_STAGES* knob: the library picks stage count, step length and ring size from the batch (64 .. 1 792 wavefronts of
    instances: either side of every threshold of the policy, several workgroups per CU sharing its LDS) - for the filter chain
    (two rows per cut) and for twelve parallel chains (thirteen rows per cut: short rings).  Round 3's sweep found wrong words
    at 98 304 instances x 8 stages (a ring shorter than the pipeline); these are the shapes it came from"""
    stages(None)
    for knob in ("FX_STAGES_GROUP", "FX_STAGES_DEBUG"):
        os.environ.pop(knob, None)
    text = progs.config2() if program == "config2" else wide_packet_program(12)
    cuts = [0, 300, 301, 345]
    x = progs.stimulus(N, cuts[-1])
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(text), b.errors()
    ys = [b.process_block(x[lo:hi]) for lo, hi in zip(cuts[:-1], cuts[1:])]
    assert b.info("kernel") >= 9
    picks = sorted(set([0, 63, 64, 511, 512, N // 2 - 1, N // 2, N - 64, N - 1] + [int(v) for v in np.linspace(0, N - 1, 300)]))
    regs = ["t", "s0", "s30", "out", "ccr"] if program == "config2" else ["a0", "a11", "s0_0", "s11_5", "out", "ccr"]
    compare_sampled(b, text, x, cuts, ys, picks, regs)
    assert b.ood_flags() == 0
    assert b.instruction_counter() == N * progs.count_instructions(text) * cuts[-1]
    print("policy: %s N=%d -> %d stage(s), LDS %d B per workgroup" % (program, N, b.info("waves_per_wg"), b.info("lds_bytes_per_wg")))


def test_the_stage_count_follows_the_planners_costs(gpu, stages):
    """round 3 chose 8 / 4 / 1 stages from the number of wavefronts alone, calibrated on the filter chain: twelve parallel chains
    (13 rows per packet) at 32 768 instances were asked for 8 stages, whose ring does not fit the LDS two workgroups share, and
    ran unstaged at 5.1e12 instead of 8.0e12 in four.  The choice now comes from the planner's costs per stage and the LDS each
    count needs (Batch::rankStages; tools/stage_policy_probe.sh measures it against pinned counts)"""
    stages(None)
    for program, N, want in (("wide12", 32768, 4), ("wide12", 98304, 2), ("config2", 4096, 8), ("config2", 262144, 1)):
        text = (progs.CONFIGS.get(program) or progs.PROBE_PROGRAMS[program])()
        b = gpu.Batch(N, 1, 0)
        assert b.load_text(text), b.errors()
        b.process_block(progs.stimulus(N, 300))
        assert b.info("waves_per_wg") == want, (program, N, b.info("waves_per_wg"))


def test_close_options_are_timed_on_the_callers_blocks(gpu, stages):
    """at 32 768 instances the cost model puts the filter chain in four and in eight stages within 20 % of each other: both are
    generated (the second one on the builder thread) and timed on the caller's own launches, three each, before one is kept
    (200-sample blocks: the 3 (K - 1) steps of filling and draining bring the plain program within reach as well - it is tried too).
    Every option computes the same words: all blocks are the oracle's, the state carries across every change of code"""
    stages(None)
    text = progs.config2()
    N, S, blocks = 32768, 200, 18   # (26 MB per block: below the size from which host blocks go through the device in overlapping pieces)
    x = progs.stimulus(N, S * blocks)
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(text), b.errors()
    ys, waves = [], []
    for k in range(blocks):
        ys.append(b.process_block(x[k * S:(k + 1) * S]))
        waves.append(b.info("waves_per_wg"))
    assert b.info("stage_trials") >= 6, (b.info("stage_trials"), waves)
    assert set(waves) >= {4, 8} and len(set(waves[-6:])) == 1, waves   # both were tried (and whatever else the model could not rule out), one was kept
    assert b.info("xlate_builds") == 1 and b.info("xlate_background_builds") >= 1
    cuts = [k * S for k in range(blocks + 1)]
    picks = [0, 63, 64, 4095, 4096, N // 2, N - 65, N - 1] + [int(v) for v in np.linspace(0, N - 1, 40)]
    compare_sampled(b, text, x, cuts, ys, sorted(set(picks)), ["t", "s0", "s30", "out", "ccr"])
    assert b.ood_flags() == 0
    assert b.instruction_counter() == N * 64 * S * blocks


def test_trials_with_launches_queued_behind_each_other(gpu, stages):
    """a caller that keeps launches queued (device-resident PCM, no sync per block: bench.py's way): the tuner only sees the time
    of a launch that has finished when the next one is made - an unfinished one is skipped, and asking for it must leave nothing
    behind that the helpers launched in between (a register fill, the counter reduction) could mistake for an error of theirs"""
    import torch
    stages(None)
    text = progs.config2()
    N, S, blocks = 32768, 200, 24
    x = progs.stimulus(N, S * blocks)
    xd = torch.from_numpy(x).cuda()
    yd = torch.empty_like(xd)
    torch.cuda.synchronize()
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(text), b.errors()
    for k in range(blocks):
        if k == 9:
            assert b.set_register("cutoff", 0.2) == 0           # (a row fill between queued launches)
        if k == 15:
            assert b.instruction_counter() == N * 64 * S * 15  # (a reduction; it waits for the launches before it)
        if k in (6, 18):
            b.sync()                                            # now and then the caller does wait: those launches can be timed
        b.process_block_dev(xd[k * S:(k + 1) * S].data_ptr(), yd[k * S:(k + 1) * S].data_ptr(), S)
    b.sync()
    y = yd.cpu().numpy()
    for n in (0, 63, 64, N // 2, N - 1):
        o = Oracle(1)
        assert o.load_text(text)
        r1 = o.process_block(x[:9 * S, n].copy())
        o.set_register("cutoff", 0.2)
        r2 = o.process_block(x[9 * S:, n].copy())
        assert np.array_equal(np.concatenate([r1, r2]).view(np.uint32), np.ascontiguousarray(y[:, n]).view(np.uint32)), n
        assert b.instruction_counter_i(n) == o.instruction_counter()
    assert b.ood_flags() == 0 and b.info("kernel") >= 9


def test_code_follows_the_block_length(gpu, stages, monkeypatch):
    """staged code is generated for a class of block lengths (long steps for long blocks, a barrier per sample and at most four
    stages for blocks of a few dozen samples); a caller that changes its block length for good gets new code after a few
    blocks - with the state carried over, bit for bit - and code that was generated before comes back at once.  (Without the
    builder thread, so that the block at which code changes is the same in every run.)"""
    stages(None)
    monkeypatch.setenv("FX_BUILDER", "0")
    N = 130
    lens = [300, 16, 16, 16, 16, 16, 16, 400, 400, 400, 400, 400, 64, 64, 64, 64, 64]
    cuts = [0]
    for n in lens:
        cuts.append(cuts[-1] + n)
    x = progs.stimulus(N, cuts[-1])
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(progs.config2())
    ys, builds, waves = [], [], []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        ys.append(b.process_block(x[lo:hi]))
        builds.append(b.info("xlate_builds"))
        waves.append(b.info("waves_per_wg"))
    # 300: built; the fourth 16-sample block: built; the first 400-sample block: the code of block 0 again (no build); the fourth 64-sample block: built
    assert builds[0] == 1 and builds[3] == 1 and builds[4] == 2 and builds[11] == 2 and builds[14] == 2 and builds[-1] == 3, builds
    assert waves[0] == 8 and waves[5] == 4 and waves[7] == 8 and waves[-1] == 8, waves
    assert b.info("code_cache_hits") == 1 and b.info("code_cached") == 3
    for n in (0, 63, 64, N - 1):
        o = Oracle(1)
        assert o.load_text(progs.config2())
        for (lo, hi), y in zip(zip(cuts[:-1], cuts[1:]), ys):
            r = o.process_block(x[lo:hi, n].copy())
            assert np.array_equal(r.view(np.uint32), np.ascontiguousarray(y[:, n]).view(np.uint32)), (n, lo, hi)
        assert b.instruction_counter_i(n) == o.instruction_counter()
    assert b.ood_flags() == 0


@pytest.mark.parametrize("builder", [True, False], ids=["builder_thread", "callers_thread_only"])
def test_returning_to_a_block_length_is_a_swap(gpu, stages, monkeypatch, builder):
    """a host that alternates between two block lengths - runs of five 16-sample blocks and five 400-sample blocks, ten times
    over (the caller the reference is written for processes 32-sample blocks, source/main.cpp:103-122, include/FX8010.h:38) -
    gets each class of code generated once: every later change is a pointer swap.  With the builder thread no translation at
    all happens on the caller's thread after the first block"""
    stages(None)
    if not builder:
        monkeypatch.setenv("FX_BUILDER", "0")
    N = 130
    lens = ([16] * 5 + [400] * 5) * 10
    cuts = [0]
    for n in lens:
        cuts.append(cuts[-1] + n)
    x = progs.stimulus(N, cuts[-1])
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(progs.config2())
    ys, waves = [], []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        ys.append(b.process_block(x[lo:hi]))
        waves.append(b.info("waves_per_wg"))
    assert b.info("xlate_builds") <= (1 if builder else 3), (b.info("xlate_builds"), b.info("xlate_background_builds"))
    assert b.info("code_cache_hits") >= (10 if builder else 17)
    if not builder:   # (the builder thread delivers when it is done, and stage counts the cost model cannot tell apart are timed first)
        assert waves[4] == 4 and waves[9] == 8, waves   # by the end of each run its own code is in force
        assert waves[10] == 4 and waves[15] == 8        # ... and from the second round on at a run's first block
        assert waves[-10] == 4 and waves[-6] == 4 and waves[-5] == 8 and waves[-1] == 8, waves
    else:
        assert waves[-10] == waves[-6] and waves[-5] == waves[-1] and waves[-1] >= waves[-6] >= 2, waves   # each class has settled on its stage count
    for n in (0, 63, 64, N - 1):
        o = Oracle(1)
        assert o.load_text(progs.config2())
        for (lo, hi), y in zip(zip(cuts[:-1], cuts[1:]), ys):
            r = o.process_block(x[lo:hi, n].copy())
            assert np.array_equal(r.view(np.uint32), np.ascontiguousarray(y[:, n]).view(np.uint32)), (n, lo, hi)
        assert b.instruction_counter_i(n) == o.instruction_counter()
        for reg in ("t", "s0", "s30", "out", "ccr"):
            assert b.get_register_bits_i(reg, n) == o.get_register_bits(reg), (reg, n)
    assert b.ood_flags() == 0


def test_small_batches_are_staged_by_default(gpu, stages):
    stages(None)
    b = gpu.Batch(300, 1, 0)
    assert b.load_text(progs.config2())
    b.process_block(progs.stimulus(300, 300))
    assert b.info("waves_per_wg") == 8          # 5 wavefronts of instances: eight stages each
    asked = gpu.Batch(300, 1, 0)
    assert asked.load_text(progs.config2())
    assert asked.info("kernel") >= 9            # a question before the first block generates code - for the shortest class of blocks
    asked.process_block(progs.stimulus(300, 300))
    assert asked.info("waves_per_wg") == 8      # ... the first block brings its own class
    short = gpu.Batch(300, 1, 0)
    assert short.load_text(progs.config2())
    short.process_block(progs.stimulus(300, 8))
    assert short.info("waves_per_wg") == 4      # ... four for blocks this short: a pipeline fills and drains in 3 (K - 1) steps
    big = gpu.Batch(262144, 1, 0)
    assert big.load_text(progs.config2())
    big.process_block(progs.stimulus(262144, 2))
    assert big.info("waves_per_wg") == 1        # 4096 wavefronts fill the machine as they are
    for name in ("config3", "config4", "config5"):
        c = gpu.Batch(300, 1, 0)
        assert c.load_text(progs.CONFIGS[name]())
        c.process_block(progs.stimulus(300, 4))
        assert c.info("waves_per_wg") == 1, name  # one recurrence from end to end: no legal cut (tests/test_stages.py)


@pytest.mark.parametrize("k", [2, 4, 8])
def test_non_finite_values_cross_the_cuts(gpu, stages, k):
    """a NaN / Inf that enters stage 0 reaches the later stages through the packets.  A stage that leaves its fast stream says
    so in its flag row (LDS), and the next stage reads that row behind every barrier - three barriers before the first such
    packet is consumed - and continues in its own exact stream: also in blocks shorter than a group of samples (the check
    behind the cold entry's barriers), in the last sample of a block, and at a block's very first sample"""
    stages(k)
    N, S = 130, 64
    x = progs.stimulus(N, S).copy()
    x[5, 3] = np.nan
    x[9, 64] = -np.inf
    x[17, 129] = np.float32(np.nan)
    x.view(np.uint32)[22, 70] = 0xFFA00123   # a signalling NaN with a payload
    x[39, 1] = np.inf                         # the last sample of a block
    x[40, 65] = np.nan                        # a block of one sample
    x[41, 66] = np.nan                        # the first sample of a block of two
    x[47, 2] = -np.nan                        # in a block of five, behind the middle
    x[63, 100] = np.nan                       # the very last sample
    run_and_compare(gpu, progs.config2(), x, [(0, 40), (40, 41), (41, 43), (43, 48), (48, S)], ["t", "s0", "s12", "s30", "out", "ccr"])


def test_delay_lines_noise_skip_and_tables_in_stages(gpu, stages):
    stages(4)
    text = ("itramsize 11 \ninput in 0\noutput out 0\ncontrol k = 0.25\nstatic noise\nstatic rd\nstatic a\nstatic t\nstatic u\nstatic w\n"
            + "".join("static s%d\n" % i for i in range(10))
            + "idelay read, rd, at, 0\nmacs a, in, rd, 0.5\nmacs a, a, noise, 0.125\nidelay write, a, at, 0\n"
            + "".join("interp s%d, s%d, k, %s\nmacs t, s%d, a, 0.05\n" % (i, i, "a" if i == 0 else "t", i) for i in range(4))
            + "macs u, t, 0, 0\nskip ccr, ccr, 6, 2\nmacs t, t, 0.5, 0.5\nmacs out, out, t, 0.1\nlog w, u, 3, 0\nexp u, w, 5, 0\n"
            + "".join("interp s%d, s%d, k, %s\nmacs t, s%d, u, 0.05\n" % (i, i, "t", i) for i in range(4, 10))
            + "macs out, out, t, 0.5\nend")
    N, S = 150, 61
    x = progs.stimulus(N, S)

    def pre(b, on):
        if b is not None:
            for n in range(N):
                b.seed_noise_i(n, 1234 + n, -77 * n)
        else:
            o, n = on
            o.seed_noise(1234 + n, -77 * n)

    b = run_and_compare(gpu, text, x, [(0, 20), (20, 21), (21, S)], ["rd", "a", "t", "u", "w", "s0", "s5", "s9", "out", "ccr"], pre=pre)
    assert b.info("waves_per_wg") >= 2


def test_stage_fuzz_slice(gpu, stages):
    stages(None)
    import fuzz_stages
    argv = sys.argv
    try:
        sys.argv = ["fuzz_stages.py", "900000", "60"]
        assert fuzz_stages.main() == 0
    finally:
        sys.argv = argv
        os.environ.pop("FX_STAGES", None)
