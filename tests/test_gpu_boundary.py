"""GPU tests of the drop-in boundary itself: the multi-device batch handle, the reference's setChannels() meaning,
loads that fail after a good one, and the reference's own console harness running on the device."""
import os
import subprocess

import numpy as np
import pytest

import fx8010_programs as progs
from pyoracle import Oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("name,n,s", [("config5", 1000, 96), ("config4", 333, 64), ("config3", 200, 1100)])
def test_two_shards_on_one_gpu_reproduce_the_single_handle(gpu, name, n, s):
    """fxb_create_on_devices with two shards (both on device 0: one host thread + stream each) against fxb_create:
    outputs, registers, counters and flags bit for bit - through host buffers (each shard copies its columns of the
    caller's [S][N] arrays) and through per-shard device buffers."""
    import torch

    text = progs.CONFIGS[name]()
    x = progs.stimulus(n, s)
    one = gpu.Batch(n, 1, 0)
    two = gpu.Batch(n, 1, devices=[0, 0])
    assert one.load_text(text) and two.load_text(text)
    sh = two.shards()
    assert len(sh) == 2 and sh[0][1] == 0 and sh[0][2] % 64 == 0 and sh[0][2] + sh[1][2] == n and sh[1][1] == sh[0][2]
    # per-instance state on both sides of the cut
    for inst, v in ((0, 0.25), (sh[0][2] - 1, -0.5), (sh[0][2], 0.125), (n - 1, 0.75)):
        key = {"config5": "u", "config4": "o", "config3": "t"}[name]
        one.set_register_i(key, inst, v)
        two.set_register_i(key, inst, v)
    y1 = one.process_block(x[: s // 2])
    y2 = two.process_block(x[: s // 2])
    assert np.array_equal(bits(y1), bits(y2))
    # second half through device-resident buffers, one pair per shard
    xs = [torch.from_numpy(np.ascontiguousarray(x[s // 2:, f:f + c])).cuda() for _, f, c in sh]
    ys = [torch.empty_like(t) for t in xs]
    torch.cuda.synchronize()
    two.process_block_dev_shards([t.data_ptr() for t in xs], [t.data_ptr() for t in ys], s - s // 2)
    two.sync()
    y1b = one.process_block(x[s // 2:])
    y2b = np.concatenate([t.cpu().numpy() for t in ys], axis=1)
    assert np.array_equal(bits(y1b), bits(y2b))
    assert one.instruction_counter() == two.instruction_counter()
    assert one.ood_flags() == two.ood_flags() == 0
    for inst in (0, sh[0][2] - 1, sh[0][2], n - 1):
        assert one.instruction_counter_i(inst) == two.instruction_counter_i(inst)
        for r in ("ccr", "out"):
            assert one.get_register_bits_i(r, inst) == two.get_register_bits_i(r, inst)
    # and against the oracle for the instances next to the cut
    for inst in (sh[0][2] - 1, sh[0][2]):
        o = Oracle(1)
        assert o.load_text(text)
        key = {"config5": "u", "config4": "o", "config3": "t"}[name]
        o.set_register(key, {sh[0][2] - 1: -0.5, sh[0][2]: 0.125}[inst])
        ref = o.process_block(x[:, inst].copy())
        got = np.concatenate([y2[:, inst], y2b[:, inst]])
        assert np.array_equal(bits(ref), bits(got)), inst


def test_sharded_arrays_and_broadcasts(gpu):
    n, s = 300, 40
    text = progs.config2()
    x = progs.stimulus(n, s)
    one, three = gpu.Batch(n, 1, 0), gpu.Batch(n, 1, devices=[0, 0, 0])
    assert one.load_text(text) and three.load_text(text)
    assert len(three.shards()) == 3
    vals = np.linspace(0.01, 0.9, n).astype(np.float32)
    for b in (one, three):
        b.set_register_array("cutoff", vals)
    assert np.array_equal(bits(one.process_block(x)), bits(three.process_block(x)))
    assert np.array_equal(bits(one.get_register_array("t")), bits(three.get_register_array("t")))
    for b in (one, three):
        b.set_register("cutoff", 0.4)
    assert np.array_equal(bits(one.process_block(x)), bits(three.process_block(x)))
    with pytest.raises(RuntimeError):
        three.process_block_dev(0, 0, 4)  # one buffer pair cannot feed three shards


def test_more_shards_than_wavefronts_is_refused(gpu):
    with pytest.raises(RuntimeError):
        gpu.Batch(3, 1, devices=[0, 0, 0, 0])


def test_set_channels_only_moves_the_loaders_bound(gpu):
    """reference: setChannels() changes numChannels, which only the loader's I/O-index check reads (FX8010.h:73,
    FX8010.cpp:447); the one-sample buffers keep their constructed size.  Here: the device layout stays, a program that
    then declares an index beyond it is refused at lowering (the reference would index out of bounds)."""
    import ctypes as C

    lib = gpu.load()
    b = gpu.Batch(70, 1, 0)
    assert not b.load_text("input in 1\noutput out 0\nmacs out, 0, in, 0.5\nend")  # index 1 > channels - 1
    assert b.errors()[1][0].startswith("I/O Index")
    h = lib.fx_create(1)
    lib.fx_set_channels(h, 2)
    assert lib.fx_get_channels(h) == 2
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "chan2.da")
    with open(path, "wb") as fh:
        fh.write(b"input in 1\noutput out 0\nmacs out, 0, in, 0.5\nend")
    assert lib.fx_load_file(h, path.encode()) == 1  # the loader accepts index 1 now
    x = (C.c_float * 1)(0.5)
    y = (C.c_float * 1)(9.0)
    assert lib.fx_process(h, x, y) == -4  # FX_E_PROGRAM: outside the parity domain, nothing was written out of bounds
    assert b"I/O index 1" in lib.fx_last_error(h)
    assert y[0] == 9.0
    lib.fx_destroy(h)
    # a batch that was constructed with two channels runs the same program
    b2 = gpu.Batch(70, 2, 0)
    assert b2.load_text("input in 1\noutput out 0\nmacs out, 0, in, 0.5\nend")
    xx = np.zeros((8, 2, 70), dtype=np.float32)
    xx[:, 0, :] = 0.25
    xx[:, 1, :] = 0.5
    # (the reference reads an X operand's input through A's channel - A is the literal 0 here, channel 0: FX8010.cpp:1058)
    assert np.all(b2.process_block(xx)[:, 0, :] == 0.125)


def test_failed_load_after_a_good_one_keeps_the_state_rows_apart(gpu):
    """a load that fails still appends its literals and declarations (as the reference does); writing such a new register
    must not land in the output-latch / cursor / LFSR / counter rows behind the old register block"""
    n, s = 130, 50
    text = progs.config3()
    x = progs.stimulus(n, s)
    b = gpu.Batch(n, 1, 0)
    assert b.load_text(text)
    y0 = b.process_block(x[:20])
    nregs = b.info("num_registers")
    assert not b.load_text("static fresh1\nstatic fresh2\nmacs fresh1, 0.123, 0.456, 0.789\nthis is not a line\nend")
    assert b.info("num_registers") > nregs
    b.set_register("fresh1", 0.5)
    b.set_register_i("fresh2", 3, 0.25)
    b.set_register_i("0.789", n - 1, 0.75)
    assert b.get_register_i("fresh2", 3) == 0.25 and b.get_register_i("fresh2", 4) == 0.0
    y1 = b.process_block(x[20:])
    bad = "static fresh1\nstatic fresh2\nmacs fresh1, 0.123, 0.456, 0.789\nthis is not a line\nend"
    for inst in (0, 3, 64, n - 1):
        # the same call sequence on the oracle: like the reference, the failed load has appended its registers AND its
        # two instructions (they run behind the first program's END from now on, FX8010.cpp:1033-1043)
        o = Oracle(1)
        assert o.load_text(text)
        r0 = o.process_block(x[:20, inst].copy())
        assert not o.load_text(bad)
        o.set_register("fresh1", 0.5)
        if inst == 3:
            o.set_register("fresh2", 0.25)
        if inst == n - 1:
            o.set_register("0.789", 0.75)
        r1 = o.process_block(x[20:, inst].copy())
        assert np.array_equal(bits(np.concatenate([r0, r1])), bits(np.concatenate([y0[:, inst], y1[:, inst]]))), inst
        assert b.instruction_counter_i(inst) == o.instruction_counter() == 256 * 20 + 258 * 30
        for r in ("fresh1", "fresh2", "0.789", "rd", "ccr"):
            assert b.get_register_bits_i(r, inst) == o.get_register_bits(r), (inst, r)
    assert b.ood_flags() == 0


HARNESS = os.path.join(ROOT, "oracle", "_ref", "main_dropin")


@pytest.mark.skipif(not os.path.exists(HARNESS), reason="oracle/_ref/main_dropin not built (needs /root/reference at build time)")
def test_the_references_own_harness_runs_on_the_device(gpu, tmp_path):
    """oracle/_ref/main_dropin = the reference's source/main.cpp + helpers.cpp compiled where they lie against
    host/FX8010.h and linked with libfx8010_amd.so (oracle/Makefile `harness`).  It loads ./testcode.da, runs its
    32-sample slider test and prints instruction count, a register, metadata and the control list - compared with what
    oracle/_ref/main_ref (the same file linked with the reference's own FX8010.cpp) prints."""
    with open(tmp_path / "testcode.da", "wb") as fh:
        fh.write(progs.config1_shipped().encode())
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "fx8010-emulator-core_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    got = subprocess.run([HARNESS], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=120)
    assert got.returncode == 0, got.stderr
    ref = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "main_ref")], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert ref.returncode == 0

    def facts(out):
        keep = []
        for line in out.splitlines():
            if "Instructions pro Audioblock" in line:
                keep.append(line.split(" fuer ")[1])  # the count, not the microseconds
            elif line.startswith(("Registerwert", "name:", "engine:", "comment:", "volume", "pan", "filter_cutoff", "Erlaubtes")):
                keep.append(line)
        return sorted(keep)

    assert facts(got.stdout) == facts(ref.stdout) and len(facts(got.stdout)) >= 8


# ---- control tracks (fxb_set_register_track)
@pytest.fixture(params=["default", "asm", "hip"], ids=["xlate", "asm", "hip"])
def track_tier(request, monkeypatch):
    monkeypatch.delenv("FX_KERNEL", raising=False)
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    if request.param != "default":
        monkeypatch.setenv("FX_KERNEL", request.param)
    return request.param


def test_slider_schedule_of_the_reference_harness_in_one_launch(gpu, track_tier):
    """source/main.cpp:103-122 sets `volume` to 0.1 / 0.25 / 0.5 / 1.0 every 8 samples around process(); tests/golden/slider.json
    is what the reference then outputs.  Here: ONE 32-sample block with the schedule as a control track."""
    import json

    with open(os.path.join(ROOT, "tests", "golden", "slider.json")) as fh:
        case = json.load(fh)[0]
    x1 = np.frombuffer(bytes.fromhex(case["input"]), dtype=np.uint32).view(np.float32)
    want = np.frombuffer(bytes.fromhex(case["output"]), dtype=np.uint32)
    N = 130
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(case["program"])
    b.set_register_track("volume", [0.1, 0.25, 0.5, 1.0], 8)
    y = b.process_block(np.repeat(x1.reshape(-1, 1), N, axis=1).copy())
    if track_tier == "default":
        assert b.info("kernel") >= 9  # the translated program applied the schedule itself
    for n in (0, 63, 64, N - 1):
        assert np.array_equal(np.ascontiguousarray(y[:, n]).view(np.uint32), want), n
        assert b.instruction_counter_i(n) == case["counter"]
        for reg, bits_ in case["registers"].items():
            assert b.get_register_bits_i(reg, n) == bits_, (reg, n)
    # the schedule was one-shot: the next block runs with the last value, like the reference object would
    y2 = b.process_block(np.repeat(x1.reshape(-1, 1), N, axis=1).copy())
    o = Oracle(1)
    assert o.load_text(case["program"])
    for lo, v in ((0, 0.1), (8, 0.25), (16, 0.5), (24, 1.0)):
        o.set_register("volume", v)
        o.process_block(x1[lo:lo + 8].copy())
    ref2 = o.process_block(x1.copy())
    assert np.array_equal(bits(ref2), bits(y2[:, 7]))


def test_tracks_per_instance_and_broadcast_together(gpu, track_tier):
    """two schedules with different periods - one value per instance and step, one value for all - inside a 100-sample block of
    config3 (TRAM reads issued a sample ahead, fp64 path), then a plain block; against the oracle doing the same with
    set_register between sub-blocks"""
    n, s = 200, 100
    text = progs.config3()
    x = progs.stimulus(n, s + 30)
    rng = np.random.default_rng(5)
    cut = rng.uniform(0.01, 0.9, size=(4, n)).astype(np.float32)       # period 32: samples 0, 32, 64, 96
    fbv = np.array([0.5, 0.25, -0.4, 0.6, 0.1, 0.45], dtype=np.float32)  # period 24: samples 0, 24, 48, 72, 96 (6th value unused)
    b = gpu.Batch(n, 1, 0)
    assert b.load_text(text)
    b.process_block(x[:10])  # state before the tracked block
    b.set_register_track("cutoff", cut, 32)
    b.set_register_track("fb", fbv, 24)
    y = b.process_block(x[10:10 + s])
    y2 = b.process_block(x[10 + s:])
    assert b.ood_flags() == 0
    for inst in (0, 63, 64, 130, n - 1):
        o = Oracle(1)
        assert o.load_text(text)
        o.process_block(x[:10, inst].copy())
        outs = []
        for lo in range(s):
            if lo % 32 == 0:
                o.set_register("cutoff", float(cut[lo // 32, inst]))
            if lo % 24 == 0:
                o.set_register("fb", float(fbv[lo // 24]))
            outs.append(o.process_block(x[10 + lo:11 + lo, inst].copy()))
        ref = np.concatenate(outs)
        assert np.array_equal(bits(ref), bits(y[:, inst])), inst
        ref2 = o.process_block(x[10 + s:, inst].copy())
        assert np.array_equal(bits(ref2), bits(y2[:, inst])), inst
        assert b.instruction_counter_i(inst) == o.instruction_counter()
        assert b.get_register_bits_i("cutoff", inst) == o.get_register_bits("cutoff")
        assert b.get_register_bits_i("fb", inst) == o.get_register_bits("fb")


def test_track_limits_and_sharded_tracks(gpu):
    n = 300
    b = gpu.Batch(n, 1, devices=[0, 0])
    assert b.load_text(progs.config5())
    names = ["damp", "decay", "diff"] + ["lp%d" % i for i in range(4)] + ["y%d" % i for i in range(9)]
    for key in names:                          # sixteen registers can have schedules ...
        b.set_register_track(key, [0.3, 0.2], 16)
    with pytest.raises(RuntimeError):
        b.set_register_track("in", [0.0], 4)   # ... a seventeenth cannot
    import ctypes as C
    one_value = (C.c_float * 1)(0.5)
    assert b._lib.fxb_set_register_track(b._h, b"nosuch", one_value, 1, 1, 0) == 1  # the reference's "not found"
    assert b._lib.fxb_set_register_track(b._h, b"damp", None, 1, 1, 0) == -3       # FX_E_ARG
    one = gpu.Batch(n, 1, 0)
    b = gpu.Batch(n, 1, devices=[0, 0])  # (a fresh one: the schedules armed above would still apply)
    assert one.load_text(progs.config5()) and b.load_text(progs.config5())
    x = progs.stimulus(n, 40)
    vals = np.linspace(0.1, 0.6, 3 * n).reshape(3, n).astype(np.float32)
    for bb in (b, one):
        bb.set_register_track("damp", vals, 16)
        bb.set_register_track("decay", [0.45, 0.2], 20)
    assert np.array_equal(bits(b.process_block(x)), bits(one.process_block(x)))
    assert np.array_equal(bits(b.get_register_array("damp")), bits(vals[2]))


def test_distinct_handles_on_concurrent_host_threads(gpu, monkeypatch, tmp_path):
    """the reference's objects share nothing (one per thread is safe); so must handles of this library: six host
    threads at once, each with batches of its own - different programs, sizes and tiers, created, loaded (lowered, translated,
    module-loaded), run and read concurrently - and a drop-in single-instance handle each"""
    import threading
    monkeypatch.delenv("FX_KERNEL", raising=False)
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    monkeypatch.delenv("FX_STAGES", raising=False)
    jobs = [("config2", 130, 40), ("config3", 200, 60), ("config4", 70, 30), ("config5", 96, 50), ("config2", 4100, 24), ("config4", 300, 17)]
    errors = []

    def work(k, name, n, s):
        try:
            text = progs.CONFIGS[name]()
            x = progs.stimulus(n, s, first_instance=1000 * k)
            for rep in range(3):
                b = gpu.Batch(n, 1, 0)
                assert b.load_text(text), b.errors()
                y1 = b.process_block(x[: s // 2])
                if rep == 1:
                    b.set_register_i("cutoff" if name != "config5" else "damp", 3, 0.5)
                y2 = b.process_block(x[s // 2:])
                for inst in (0, 3, 63, n - 1):
                    o = Oracle(1)
                    assert o.load_text(text)
                    r1 = o.process_block(x[: s // 2, inst].copy())
                    if rep == 1 and inst == 3:
                        o.set_register("cutoff" if name != "config5" else "damp", 0.5)
                    r2 = o.process_block(x[s // 2:, inst].copy())
                    assert np.array_equal(bits(r1), bits(y1[:, inst])) and np.array_equal(bits(r2), bits(y2[:, inst])), (name, rep, inst)
                    assert b.instruction_counter_i(inst) == o.instruction_counter()
            one = gpu.Single(1)
            path = tmp_path / ("thread%d.da" % k)
            path.write_bytes(text.encode())
            assert one.load_file(str(path))
            o = Oracle(1)
            assert o.load_text(text)
            for t in range(6):
                got = one.process([float(x[t, 0])])
                assert np.array_equal(bits(got), bits(o.process_block(x[t:t + 1, 0].copy()))), (name, t)
        except Exception as e:  # noqa: BLE001 - collected and re-raised on the main thread
            import traceback
            errors.append("%s: %s\n%s" % (name, e, traceback.format_exc()))

    threads = [threading.Thread(target=work, args=(k,) + job) for k, job in enumerate(jobs)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[0]


def test_many_tracks_in_one_block(gpu, track_tier):
    """seven schedules in one 90-sample block - periods 1 .. 40, per instance and for all, several changes due at the same
    sample - as ONE list of events the generated loop walks; against the oracle with set_register in between"""
    n, s = 150, 90
    text = progs.config5()
    x = progs.stimulus(n, s + 20)
    rng = np.random.default_rng(77)
    plan = [("damp", 8, False), ("decay", 40, False), ("diff", 1, False), ("lp1", 16, True), ("y3", 5, True), ("w2", 8, False), ("y17", 30, True)]
    sched = {}
    for name, period, per in plan:
        steps = (s + period - 1) // period + (1 if name == "decay" else 0)   # (one value more than the block uses)
        sched[name] = rng.uniform(-0.9, 0.9, size=(steps, n) if per else (steps,)).astype(np.float32)
    b = gpu.Batch(n, 1, 0)
    assert b.load_text(text)
    b.process_block(x[:7])
    for name, period, per in plan:
        b.set_register_track(name, sched[name], period)
    y = b.process_block(x[7:7 + s])
    y2 = b.process_block(x[7 + s:])
    assert b.ood_flags() == 0
    for inst in (0, 63, 64, n - 1):
        o = Oracle(1)
        assert o.load_text(text)
        o.process_block(x[:7, inst].copy())
        outs = []
        for lo in range(s):
            for name, period, per in plan:
                if lo % period == 0:
                    v = sched[name][lo // period]
                    o.set_register(name, float(v[inst] if per else v))
            outs.append(o.process_block(x[7 + lo:8 + lo, inst].copy()))
        assert np.array_equal(bits(np.concatenate(outs)), bits(y[:, inst])), inst
        assert np.array_equal(bits(o.process_block(x[7 + s:, inst].copy())), bits(y2[:, inst])), inst
        assert b.instruction_counter_i(inst) == o.instruction_counter()
        for name, _, _ in plan:
            assert b.get_register_bits_i(name, inst) == o.get_register_bits(name), (name, inst)


@pytest.mark.parametrize("channels", [1, 2])
def test_small_host_blocks_take_the_pinned_path_and_large_ones_the_staged_copies(gpu, channels):
    """Host blocks of up to 2 KB of PCM are read and written by the kernel in pinned host memory (fx_batch.cpp processHost);
    larger ones are copied.  The same instances fed in blocks of both sizes, interleaved, must follow the oracle bit for bit."""
    if channels == 1:
        text = progs.config3()
    else:
        text = "input l 0\ninput r 1\noutput ol 0\noutput or 1\nstatic a\nitramsize 40 \nidelay read, a, at, 0\nmacs ol, l, a, 0.5\nmacs or, r, ol, 0.25\nidelay write, ol, at, 0\nend"
    n = 6
    cuts = [1, 1, 40, 1, 85, 3, 200, 1, 1]          # 6 instances x {1, 3, 40} samples: pinned; x {85, 200}: staged (mono)
    s_total = sum(cuts)
    rng = np.random.default_rng(77)
    x = rng.uniform(-0.9, 0.9, size=(s_total, n) if channels == 1 else (s_total, channels, n)).astype(np.float32)
    b = gpu.Batch(n, channels, 0)
    assert b.load_text(text), b.errors()
    at, parts = 0, []
    for c in cuts:
        parts.append(b.process_block(np.ascontiguousarray(x[at:at + c])))
        at += c
    y = np.concatenate(parts, axis=0)
    for inst in range(n):
        o = Oracle(channels)
        assert o.load_text(text)
        xin = x[:, inst] if channels == 1 else np.ascontiguousarray(x[:, :, inst])
        ref = o.process_block(xin.copy())
        got = y[:, inst] if channels == 1 else y[:, :, inst]
        assert np.array_equal(ref.view(np.uint32), np.ascontiguousarray(got).view(np.uint32)), inst
        assert b.instruction_counter_i(inst) == o.instruction_counter()


def _replay_with_schedules(text, x_col, plans, inst):
    """the oracle doing what the schedules do: set_register between its process() calls"""
    o = Oracle(1)
    assert o.load_text(text)
    ref = np.empty(x_col.shape[0], dtype=np.float32)
    for t in range(x_col.shape[0]):
        for name, period, vals in plans:
            if t % period == 0 and t // period < vals.shape[0]:
                v = vals[t // period]
                o.set_register(name, float(v if vals.ndim == 1 else v[inst]))
        ref[t] = o.process_block(x_col[t:t + 1].copy())[0]
    return ref, o


def test_track_on_a_register_the_program_loads_from_a_delay_line_first(gpu, track_tier):
    """The scheduled value is set at the head of the sample, BEFORE the first instruction - here a delay-line read into the same
    register, which therefore wins.  The translated tier issues leading delay-line reads a sample ahead: not into a register
    with a schedule (api fuzz seed 50788)."""
    text = ("input in 0\noutput out 0\nstatic noise\nxtramsize 23 \nstatic r0\nstatic r2\nstatic r3\nstatic r6\nstatic r12\n"
            "xdelay read, r0, at, 0\nacc3 r3, r0, noise, r12\ninterp r2, r6, r3, 0.125\nxdelay write, in, at, 0\nmacs out, out, r2, 0.5\nend")
    N, S = 70, 60
    x = progs.stimulus(N, S)
    vals = np.random.default_rng(1).uniform(-1, 1, size=(25, N)).astype(np.float32)
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(text), b.errors()
    b.set_register_track("r0", vals, 1)
    y = b.process_block(x)
    for inst in (0, 63, 64, N - 1):
        ref, o = _replay_with_schedules(text, x[:, inst], [("r0", 1, vals)], inst)
        assert np.array_equal(bits(ref), bits(y[:, inst])), inst
        assert b.get_register_bits_i("r0", inst) == o.get_register_bits("r0")


def test_schedules_in_small_host_blocks_on_the_tiers_that_cut_the_block(gpu, track_tier):
    """Host blocks of a few samples take the pinned-memory path (no event pair around the launch); the interpreter and HIP tiers
    apply schedules by cutting the block and must still wait for each piece before they write the next step's values
    (api fuzz seed 50790: the step at sample 1 was overwritten by the epilogue of the launch before it)."""
    text = "input in 0\noutput out 0\ncontrol c = 0.5\nstatic r0\nstatic r1\nmacs r0, r0, in, 0.25\nmacs r1, r0, in, c\nmacs out, r1, in, c\nend"
    N = 64
    x = progs.stimulus(N, 16)
    vals = np.random.default_rng(3).uniform(-1, 1, size=(4, N)).astype(np.float32)
    for r1_period in (3, 2, 1):
        for S in (2, 3, 7):
            b = gpu.Batch(N, 1, 0)
            assert b.load_text(text), b.errors()
            b.set_register_track("c", vals, 1)
            b.set_register_track("r1", np.array([0.25], dtype=np.float32), r1_period)
            y = b.process_block(x[:S])
            for inst in (0, 33, 63):
                ref, o = _replay_with_schedules(text, x[:S, inst], [("c", 1, vals), ("r1", r1_period, np.array([0.25], dtype=np.float32))], inst)
                assert np.array_equal(bits(ref), bits(y[:, inst])), (r1_period, S, inst)
                assert b.get_register_bits_i("c", inst) == o.get_register_bits("c"), (r1_period, S, inst)


@pytest.mark.parametrize("shards", [1, 2])
def test_large_host_block_in_pinned_buffers_is_processed_in_overlapping_pieces(gpu, shards):
    """fxb_process_block on large host blocks: pageable caller buffers go through staged copies in overlapping pieces on three
    streams (fx_batch.cpp processHostPipelined: 42 MB each way = five pieces), pinned buffers of a single-shard handle are
    processed in place (no copies), the shards of a multi-shard handle copy their columns of the caller's rows (2-D copies, in
    pieces).  All of them are consecutive blocks to the kernel: state, delay lines and counters must come out exactly alike, and
    as the oracle says."""
    torch = pytest.importorskip("torch")
    n, s = 65536, 160 * shards                      # 42 MB each way per shard (a shard copies its columns: 2-D copies)
    text = progs.config3()
    x = progs.stimulus(n, s)
    pin_in = torch.empty(x.shape, dtype=torch.float32).pin_memory()
    pin_in.numpy()[...] = x
    pin_out = torch.empty(x.shape, dtype=torch.float32).pin_memory()
    a = gpu.Batch(n, 1, 0)
    b = gpu.Batch(n, 1, 0) if shards == 1 else gpu.Batch(n, 1, devices=[0] * shards)
    assert a.load_text(text) and b.load_text(text)
    ya = a.process_block(x)                         # pageable: one piece
    yb = b.process_block(pin_in.numpy(), pin_out.numpy())
    assert np.array_equal(ya.view(np.uint32), yb.view(np.uint32))
    assert a.instruction_counter() == b.instruction_counter()
    ya = a.process_block(x[:40])                    # state after the big block is the same
    yb = b.process_block(x[:40])
    assert np.array_equal(ya.view(np.uint32), yb.view(np.uint32))
    for inst in (0, 63, 64, n - 1):
        o = Oracle(1)
        assert o.load_text(text)
        o.process_block(x[:, inst].copy())
        ref = o.process_block(x[:40, inst].copy())
        assert np.array_equal(bits(ref), bits(yb[:, inst])), inst
        assert b.instruction_counter_i(inst) == o.instruction_counter()


def test_delay_memory_positions_and_lfsr_against_the_oracle(gpu):
    """the state a caller cannot see through outputs (reference include/FX8010.h:210-217, 290-291), read back and compared with
    the oracle's directly: config5's delay line word for word after it has wrapped (2 100 samples: first read-back at 2048),
    config3's 1000-slot line after two wraps, the four positions, the LFSR words of a noise program"""
    for name, S, which, words in (("config5", 2100, 1, 8192), ("config3", 2300, 0, 1000)):
        text = progs.CONFIGS[name]()
        N = 70
        x = progs.stimulus(N, S)
        b = gpu.Batch(N, 1, 0)
        assert b.load_text(text), b.errors()
        cuts = [0, 600, 601, 2047, S]
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            b.process_block(x[lo:hi])
        for n in (0, 63, 64, N - 1):
            o = Oracle(1)
            assert o.load_text(text)
            o.process_block(x[:, n].copy())
            assert np.array_equal(bits(b.get_tram_i(which, n, words)), bits(o.tram(which, words))), (name, n)
            assert b.get_cursors_i(n) == o.cursors(), (name, n)
            assert not np.any(b.get_tram_i(1 - which, n, 16))   # the other delay memory: untouched
    text = ("itramsize 6 \nstatic noise\nstatic rd\nstatic a\ninput in 0\noutput out 0\nidelay read, rd, at, 0\nmacs a, in, noise, 0.5\n"
            "idelay write, a, at, 2\nmacs out, rd, noise, 0.25\nend")
    N, S = 130, 77
    x = progs.stimulus(N, S)
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(text), b.errors()
    for n in range(N):
        b.seed_noise_i(n, 99 + n, -5 * n)
    b.process_block(x[:40])
    b.process_block(x[40:])
    img = b.save_state()
    hdr = np.frombuffer(img[:64].tobytes(), dtype=np.int32)
    rows = img[64:64 + int(hdr[6]) * N * 4].view(np.uint32).reshape(int(hdr[6]), N)   # (state rows of the image: [row][instance])
    for n in (0, 64, N - 1):
        o = Oracle(1)
        assert o.load_text(text)
        o.seed_noise(99 + n, -5 * n)
        o.process_block(x[:, n].copy())
        assert np.array_equal(bits(b.get_tram_i(0, n, 8192)), bits(o.tram(0, 8192))), n   # (writes at position + 2: beyond itramsize, inside the reference's array)
        assert b.get_cursors_i(n) == o.cursors()
        nregs = int(hdr[5])
        lfsr = [int(v) for v in rows[nregs + 1 + 4: nregs + 1 + 6, n].view(np.int32)]     # latches (1 channel), 4 positions, then g_x1, g_x2
        assert lfsr == o.lfsr(), n


@pytest.mark.parametrize("name", ["config5", "config4", "config2"])
def test_state_snapshot_round_trip_and_repartition(gpu, name):
    """fxb_save_state / fxb_load_state: a batch is stopped in the middle of a run, its image loaded into a NEW handle - one
    shard, and three shards (an image is laid out by global instance: a sharded handle can be re-partitioned) - and both carry
    on exactly like the batch that was never stopped: outputs, registers, counters, delay memory; per-instance register values
    and a moved control travel with the image"""
    text = progs.CONFIGS[name]()
    N, S = 333, 2200 if name == "config5" else 120
    x = progs.stimulus(N, S + 60)
    key = {"config5": "u", "config4": "o", "config2": "s7"}[name]
    ctl = {"config5": "decay", "config4": "cutoff", "config2": "cutoff"}[name]
    a = gpu.Batch(N, 1, 0)
    assert a.load_text(text), a.errors()
    a.process_block(x[:S // 2])
    a.set_register_i(key, 200, 0.3125)
    a.set_register(ctl, 0.4)
    a.process_block(x[S // 2:S])
    img = a.save_state()
    ya = a.process_block(x[S:])
    one = gpu.Batch(N, 1, 0)
    three = gpu.Batch(N, 1, devices=[0, 0, 0])
    for h in (one, three):
        assert h.load_text(text)
        h.load_state(img)
        assert h.get_register_i(ctl, 5) == np.float32(0.4)
        y = h.process_block(x[S:])
        assert np.array_equal(bits(y), bits(ya)), name
        assert h.instruction_counter() == a.instruction_counter()
        for n in (0, 63, 64, 200, N - 1):
            assert h.instruction_counter_i(n) == a.instruction_counter_i(n)
            for r in (key, ctl, "out", "ccr"):
                assert h.get_register_bits_i(r, n) == a.get_register_bits_i(r, n), (r, n)
            if name == "config5":
                assert np.array_equal(bits(h.get_tram_i(1, n, 8192)), bits(a.get_tram_i(1, n, 8192)))
                assert h.get_cursors_i(n) == a.get_cursors_i(n)
        assert h.ood_flags() == 0
    # ... and the image of the re-partitioned handle is the image of the single one
    assert np.array_equal(one.save_state(), three.save_state())
    # against the oracle: the instance that was given a value of its own
    o = Oracle(1)
    assert o.load_text(text)
    o.process_block(x[:S // 2, 200].copy())
    o.set_register(key, 0.3125)
    o.set_register(ctl, 0.4)
    ref = o.process_block(x[S // 2:, 200].copy())[S - S // 2:]
    assert np.array_equal(bits(ref), bits(ya[:, 200]))
    # an image of another program or another batch size is refused
    other = gpu.Batch(N, 1, 0)
    assert other.load_text(progs.config3())
    with pytest.raises(RuntimeError):
        other.load_state(img)
    small = gpu.Batch(N - 1, 1, 0)
    assert small.load_text(text)
    with pytest.raises(RuntimeError):
        small.load_state(img)


@pytest.mark.parametrize("tier", ["asm", "asm_lds", 1, 2, 4])
def test_snapshot_and_delay_memory_on_the_other_tiers(gpu, monkeypatch, tier):
    """the state image and the delay-memory reads do not depend on the tier that runs the program: the interpreter (VGPR and LDS
    register file) and the HIP C++ kernel with 1 / 2 / 4 instances per lane, whose delay memory is tiled [wavefront][slot][64 x K] -
    an image saved under one tier loads under another"""
    monkeypatch.delenv("FX_KERNEL", raising=False)
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    if isinstance(tier, str):
        monkeypatch.setenv("FX_KERNEL", tier)
    else:
        monkeypatch.setenv("FX_INST_PER_LANE", str(tier))
    text = ("itramsize 37 \nxtramsize 100 \nstatic rd\nstatic xd\nstatic a\nstatic noise\ninput in 0\noutput out 0\ncontrol fb = 0.5\n"
            "idelay read, rd, at, 0\nxdelay read, xd, at, 0\nmacs a, in, rd, fb\nmacs a, a, noise, 0.0625\nidelay write, a, at, 0\n"
            "interp out, out, 0.25, xd\nxdelay write, a, at, 3\nend")
    N, S = 200, 150
    x = progs.stimulus(N, S + 40)
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(text), b.errors()
    b.set_register_i("fb", 77, 0.25)
    b.process_block(x[:S])
    img = b.save_state()
    for n in (0, 63, 64, 77, 130, N - 1):
        o = Oracle(1)
        assert o.load_text(text)
        if n == 77:
            o.set_register("fb", 0.25)
        o.process_block(x[:S, n].copy())
        assert np.array_equal(bits(b.get_tram_i(0, n, 64)), bits(o.tram(0, 64))), (tier, n)
        assert np.array_equal(bits(b.get_tram_i(1, n, 128)), bits(o.tram(1, 128))), (tier, n)
        assert b.get_cursors_i(n) == o.cursors(), (tier, n)
    ya = b.process_block(x[S:])
    monkeypatch.delenv("FX_KERNEL", raising=False)       # the image goes to a handle on the default tier
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    c = gpu.Batch(N, 1, 0)
    assert c.load_text(text)
    c.load_state(img)
    assert np.array_equal(bits(c.process_block(x[S:])), bits(ya)), tier
    assert c.get_register_i("fb", 77) == np.float32(0.25) and c.get_register_i("fb", 78) == np.float32(0.5)
    assert np.array_equal(c.save_state(), b.save_state())


def test_prepare_moves_the_translation_out_of_the_first_block(gpu, monkeypatch):
    """fxb_prepare(h, n_samples, wait): the code for a stream of n_samples-sample blocks is generated when the caller says so -
    after loading, before the stream starts - and the follow-ups of a first build are waited for: the first block and the first
    touch of a slider then find their code (a real-time caller's deadline: 667 us per 32-sample block, reference
    include/FX8010.h:38)"""
    monkeypatch.delenv("FX_KERNEL", raising=False)
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    monkeypatch.delenv("FX_STAGES", raising=False)
    monkeypatch.delenv("FX_BUILDER", raising=False)
    monkeypatch.delenv("FX_STAGES_TUNE", raising=False)
    text = progs.config2()
    N = 300
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(text), b.errors()
    b.prepare(32)
    builds, background = b.info("xlate_builds"), b.info("xlate_background_builds")
    assert builds == 1 and background >= 1       # the code of 32-sample blocks; the control-row variant (and stage counts on trial) behind it
    x = progs.stimulus(N, 32 * 12)
    o = Oracle(1)
    assert o.load_text(text)
    for k in range(12):
        if k == 5:
            b.set_register("cutoff", 0.25)       # the first touch of a slider
            o.set_register("cutoff", 0.25)
        y = b.process_block(x[32 * k:32 * (k + 1)])
        assert np.array_equal(bits(o.process_block(x[32 * k:32 * (k + 1), 7].copy())), bits(y[:, 7])), k
    assert b.info("xlate_builds") == 1           # nothing was translated on the caller's thread after prepare
    with pytest.raises(RuntimeError):
        gpu.Batch(N, 1, 0).prepare(32)           # nothing loaded


def test_bench_through_rccl_with_one_rank(gpu, tmp_path):
    """the multi-GPU launch of bench.py as the driver makes it (torch.distributed.run, backend nccl = RCCL, barriers and the
    MAX / SUM / all-gather of times and counters on the device) - with ONE rank, which is what a one-GPU box can run: the
    process-group code path that the 2 / 4 / 8-GPU runs take, not the single-process shortcut"""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FX_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("FX_KERNEL", "FX_INST_PER_LANE", "FX_STAGES", "FX_BENCH_REHEARSAL"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29533",
                        os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--config", "config3", "--instances", "4096", "--samples", "256",
                        "--parity-instances", "8", "--cpu-seconds", "0", "--no-extras"], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.split("\n") if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["parity"]["parity_ok"] and d["value"] > 0
    assert len(d["roofline"]["per_gpu_kernel_ms"]) == 1 and d["roofline"]["per_gpu_kernel_ms"][0] > 0


@pytest.mark.parametrize("scaling", ["strong", "weak"])
def test_bench_splits_the_job_over_two_ranks(gpu, scaling):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, two ranks), rehearsed on the one GPU of the test box
    (FX_BENCH_REHEARSAL=1: the ranks share the device and meet over gloo - RCCL refuses two ranks on one device; same code path
    otherwise).  strong (the default): the instance count is the WHOLE job - 4 100 instances are 2 050 per rank, rank 1's stimulus
    starts at instance 2 050, the line totals 4 100; weak: 4 100 per rank, 8 200 in all.  Parity of sampled instances inside the
    run (rank 0's range) and the exact instruction total of the job (no SKIP in config3: every instance executes all 256)."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FX_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("FX_KERNEL", "FX_INST_PER_LANE", "FX_STAGES", "FX_FORCE_DIST"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
                        os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "config3", "--instances", "4100", "--samples", "128",
                        "--scaling", scaling, "--parity-instances", "8", "--cpu-seconds", "0", "--no-extras"], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.split("\n") if l.startswith("{")][-1])
    total = 4100 if scaling == "strong" else 8200
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["parity"]["parity_ok"]
    assert d["config"]["instances_total"] == total and d["config"]["per_gpu_instances"] == [total // 2, total // 2]
    assert len(d["roofline"]["per_gpu_kernel_ms"]) == 2 and min(d["roofline"]["per_gpu_kernel_ms"]) > 0
    assert abs(d["config"]["instr_per_sample_executed"] - 256.0) < 1e-9
    # value = executed instructions of BOTH ranks / the slowest rank's time
    assert abs(d["value"] * 1e6 * d["ms_per_step"] * 1e-3 * 2 - 256.0 * 128 * total * 2) / (256.0 * 128 * total * 2) < 1e-3


@pytest.mark.gpu
def test_tier_note_says_which_tier_runs_and_why(gpu, monkeypatch):
    """fxb_tier_note: the tier in force in words - and, below the translated tier, the reason: a host that finds
    FXB_INFO_KERNEL < 9 can tell its user why (a SKIP that can jump over END is the reference's own multi-pass corner,
    FX8010.cpp:1033,1243)"""
    monkeypatch.delenv("FX_KERNEL", raising=False)
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    monkeypatch.delenv("FX_STAGES", raising=False)
    x = progs.stimulus(70, 12)
    b = gpu.Batch(70, 1, 0)
    assert b.tier_note() == "no program loaded"
    assert b.load_text(progs.CONFIGS["config2"]())
    assert "not lowered yet" in b.tier_note()
    b.process_block(x)
    assert b.info("kernel") >= 9 and b.tier_note().startswith("translated to gfx950 code (fx_xlate_v"), b.tier_note()
    if b.info("waves_per_wg") > 1:
        assert "%d stages" % b.info("waves_per_wg") in b.tier_note()
    multipass = gpu.Batch(70, 1, 0)
    assert multipass.load_text("input in 0\noutput out 0\nstatic a\nmacs a, in, 0, 0\nmacs out, a, 0, 0\nskip ccr, ccr, 6, 1\nend")
    multipass.process_block(x)
    assert 2 <= multipass.info("kernel") <= 8 and multipass.tier_note().startswith("interpreter (fx_interp_v") and "END can be skipped" in multipass.tier_note(), multipass.tier_note()
    generic = gpu.Batch(70, 1, 0)
    assert generic.load_text("input in 0\noutput out 0\nstatic a\nlog a, in, 40, 0\nmacs out, a, 0, 0\nend")   # table 40: outside the reference's table vector
    generic.process_block(x)
    assert generic.info("kernel") == 0 and generic.tier_note().startswith("HIP C++ kernel (") and "out-of-range table" in generic.tier_note(), generic.tier_note()
    monkeypatch.setenv("FX_KERNEL", "asm")
    interp = gpu.Batch(70, 1, 0)
    assert interp.load_text(progs.CONFIGS["config2"]())
    interp.process_block(x)
    assert 2 <= interp.info("kernel") <= 8 and interp.tier_note().startswith("interpreter (fx_interp_v"), interp.tier_note()


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["default", "asm"])
def test_wavefronts_of_a_simd_take_turns(gpu, kernel, monkeypatch):
    """generated code gives the wavefronts of a SIMD the top priority by turns (by the clock) wherever a SIMD holds two or more -
    131 072 instances: 2 048 wavefronts on 1 024 SIMDs - and not for a single wavefront per SIMD or less; results are the same
    bits either way (priorities change who issues when, nothing else), here against the
    oracle on sampled instances and against the same batch with the mode off"""
    monkeypatch.delenv("FX_KERNEL", raising=False)
    monkeypatch.delenv("FX_INST_PER_LANE", raising=False)
    monkeypatch.delenv("FX_XLATE_PRIO", raising=False)
    if kernel != "default":
        monkeypatch.setenv("FX_KERNEL", kernel)   # (the interpreter's end-of-sample handler does the same, fx_interp_gfx950.S)
    import torch
    text = progs.CONFIGS["config3"]()
    S = 40
    small = gpu.Batch(4096, 1, 0)
    assert small.load_text(text)
    small.process_block(progs.stimulus(4096, 8))
    assert "by turns" not in small.tier_note(), small.tier_note()
    N = 131072
    x = torch.empty((S, N), dtype=torch.float32, device="cuda").uniform_(-0.9, 0.9)
    outs = []
    for mode in (None, "0"):
        if mode is None:
            monkeypatch.delenv("FX_XLATE_PRIO", raising=False)
        else:
            monkeypatch.setenv("FX_XLATE_PRIO", mode)
        b = gpu.Batch(N, 1, 0)
        assert b.load_text(text), b.errors()
        y = torch.empty_like(x)
        for at in (0, 17):   # two blocks: state carried over
            n = 17 if at == 0 else S - 17
            b.process_block_dev(x[at:at + n].data_ptr(), y[at:at + n].data_ptr(), n)
        b.sync()
        if kernel == "default":
            assert b.info("kernel") >= 9 and (("by turns" in b.tier_note()) == (mode is None)), b.tier_note()
        else:
            assert 2 <= b.info("kernel") <= 8
        outs.append(y.cpu().numpy())
        if mode is None:
            xh = x.cpu().numpy()
            for inst in (0, 63, 64, 65535, 99999, N - 1):
                o = Oracle(1)
                assert o.load_text(text)
                ref = o.process_block(xh[:, inst].copy())
                assert np.array_equal(ref.view(np.uint32), np.ascontiguousarray(outs[0][:, inst]).view(np.uint32)), inst
                assert b.instruction_counter_i(inst) == o.instruction_counter()
        del b
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))


RELEASE_KNOB_VALUES = [
    ("FX_KERNEL", ["xlate", "xlate_v168", "asm", "asm_v128", "asm_lds", "hip"]),
    ("FX_INST_PER_LANE", ["1", "2", "4"]),
    ("FX_STAGES", ["1", "2", "4", "8", "16"]),
    ("FX_STAGES_GROUP", ["1", "2", "4"]),
    ("FX_STAGES_TUNE", ["0"]),
    ("FX_BUILDER", ["0"]),
    ("FX_XLATE_PRIO", ["0", "1"]),
    ("FX_HOST_PIPELINE", ["0"]),
]


def test_no_release_knob_changes_a_bit(gpu, monkeypatch):
    """INTEGRATION.md's table of environment knobs - everything the release library reads (csrc/fx_knobs.hpp; the CPU half:
    tests/test_release_knobs.py) - value by value on config3 (delay line with feedback, 256 instructions): three blocks through
    the host boundary, one of them large enough (33 MiB) for the pipelined copy that FX_HOST_PIPELINE switches off, then every
    output word, the registers and delay-line positions of sampled instances, every instance's instruction counter total and
    the out-of-domain flags must equal the default's; the default itself is checked against the oracle.  A knob may cost time,
    never a bit: the contract is the reference's process() (source/FX8010.cpp:1023-1249)."""
    for k, _ in RELEASE_KNOB_VALUES:
        monkeypatch.delenv(k, raising=False)
    text = progs.CONFIGS["config3"]()
    N, cuts = 4100, [0, 7, 39, 39 + 2048]
    x = progs.stimulus(N, cuts[-1])
    watch = (0, 63, 64, 2049, N - 1)

    def run():
        b = gpu.Batch(N, 1, 0)
        assert b.load_text(text), b.errors()
        b.set_register_i("t", 64, 0.125)     # one per-instance value: a row that the code did not ask for
        y = np.concatenate([b.process_block(x[lo:hi]) for lo, hi in zip(cuts[:-1], cuts[1:])], axis=0)
        regs = {n: [b.get_register_bits_i(r, n) for r in ("t", "a", "rd", "out", "ccr")] + b.get_cursors_i(n) + [b.instruction_counter_i(n)] for n in watch}
        return y, regs, b.instruction_counter(), b.ood_flags(), b.tier_note()

    y0, regs0, count0, ood0, note0 = run()
    assert ood0 == 0 and note0.startswith("translated to gfx950 code")
    for n in watch:
        o = Oracle(1)
        assert o.load_text(text)
        if n == 64:
            o.set_register("t", 0.125)
        ref = o.process_block(x[:, n].copy())
        assert np.array_equal(bits(ref), bits(y0[:, n])), n
        assert regs0[n][-1] == o.instruction_counter() and regs0[n][5:9] == o.cursors()
    seen = set()
    for knob, values in RELEASE_KNOB_VALUES:
        for v in values:
            monkeypatch.setenv(knob, v)
            if knob == "FX_STAGES_GROUP":
                monkeypatch.setenv("FX_STAGES", "4")      # (a ring length only means something for a staged program)
            y, regs, count, ood, note = run()
            monkeypatch.delenv(knob)
            monkeypatch.delenv("FX_STAGES", raising=False)
            seen.add(note.split(":")[0].split(",")[0])
            assert np.array_equal(bits(y), bits(y0)), "%s=%s changed an output word (%s)" % (knob, v, note)
            assert regs == regs0 and count == count0 and ood == ood0, "%s=%s (%s)" % (knob, v, note)
    assert len(seen) >= 4, seen   # the knobs did reach other tiers and builds


@pytest.mark.gpu
@pytest.mark.parametrize("n", [300, 20000], ids=["staged", "plain"])
def test_controls_that_stay_put_go_back_into_the_code(gpu, n, monkeypatch):
    """A host moves ONE slider of config5's three (the reference's harness: setRegisterValue("volume", ...) between blocks,
    /root/reference/source/main.cpp:107-114).  The first touch switches to code with every declared control in a row (built
    ahead: a pointer swap); a few blocks later the controls that have not moved are folded back into the code (the lean variant,
    from the builder thread: a constant X makes an INTERP 16 issue clocks cheaper than a row) - and the full variant comes back
    at once when a second slider starts moving, when a cold control gets per-instance values or a schedule.  Controls that are
    left alone for 8192 sample periods cool down and are folded in again.  Whatever variant runs, every word equals the
    oracle's for the same calls; no control variant is ever translated on the caller's thread."""
    for k in ("FX_KERNEL", "FX_INST_PER_LANE", "FX_BUILDER", "FX_STAGES"):
        monkeypatch.delenv(k, raising=False)
    # (config5 with its output gain in a register of its own, `g`, that no instruction writes: folded into the code like a control)
    text, S = progs.config5().replace("static u\n", "static u\nstatic g = 0.5\n").replace("macs out, 0, m, 0.5", "macs out, 0, m, g"), 32
    assert "static g = 0.5" in text and "macs out, 0, m, g" in text
    b = gpu.Batch(n, 1, 0)
    assert b.load_text(text), b.errors()
    blocks = 48
    x = progs.stimulus(n, S * blocks).reshape(blocks, S, n)
    watch = sorted({0, 5, 63, 64, n // 2, n - 1})
    oracles = {}
    for i in watch:
        oracles[i] = Oracle(1)
        assert oracles[i].load_text(text)

    def both(name, v, inst=None):
        if inst is None:
            b.set_register(name, v)
            for o in oracles.values():
                o.set_register(name, v)
        else:
            b.set_register_i(name, inst, v)
            if inst in oracles:
                oracles[inst].set_register(name, v)

    rows, k = [], 0
    def step(times=1, settle=False):
        nonlocal k
        for _ in range(times):
            if settle:
                b.prepare(S, True)      # (the builder thread has finished what it was asked for: the next block can adopt it)
            y = b.process_block(x[k].copy())
            for i, o in oracles.items():
                ref = o.process_block(x[k][:, i].copy())
                assert np.array_equal(bits(ref), bits(y[:, i])), (k, i, b.info("control_rows"))
            rows.append(b.info("control_rows"))
            k += 1

    b.prepare(S, True)
    step(2)
    assert rows[-1] == 0
    both("decay", 0.4)
    step()
    assert rows[-1] == 3                                   # the whole panel at the first touch
    step(3, settle=True)
    assert rows[-1] == 1                                   # ... then only the slider that moves
    for v in (0.1, 0.25, 0.5, 1.0, 0.3):                   # (the harness's values, main.cpp:80)
        both("decay", v)
        step()
    assert rows[-1] == 1
    # another register gets per-instance values while the lean code runs: it needs a row of its own in the NEXT block's code (the
    # lean variant remembers which controls it folded, not which registers had rows: the API fuzzer's control panel found that)
    held = b.info("num_rows")
    both("g", 0.25, inst=5)
    step()
    assert b.info("num_rows") == held + 1 and rows[-1] == 1
    step(2, settle=True)
    both("damp", 0.2)                                      # a second slider: its value is folded into the lean code
    step()
    assert rows[-1] == 3
    step(3, settle=True)
    assert rows[-1] == 2
    both("diff", 0.55, inst=5)                             # per-instance values for the control that never moved
    step()
    assert rows[-1] == 3
    step(3, settle=True)
    both("diff", 0.6)                                      # levelled again by a broadcast write: it has moved now, the row stays
    both("decay", 0.45)
    step(2)
    # a schedule inside a block on a control (applied by the generated loop itself)
    b.set_register_track("damp", np.array([0.3, 0.1], dtype=np.float32), 16)
    y = b.process_block(x[k].copy())
    for i, o in oracles.items():
        o.set_register("damp", 0.3)
        r0 = o.process_block(x[k][:16, i].copy())
        o.set_register("damp", 0.1)
        r1 = o.process_block(x[k][16:, i].copy())
        assert np.array_equal(bits(np.concatenate([r0, r1])), bits(y[:, i])), i
    k += 1
    step(2)
    # controls that are left alone cool down (8192 sample periods without a write): one long block, then their values are folded
    # in again (`diff` was levelled by its broadcast write) - and the first touch after the rest is a swap once more
    rest = progs.stimulus(n, 8192 + 64, first_sample=S * blocks)
    y = b.process_block(rest)
    for i, o in oracles.items():
        assert np.array_equal(bits(o.process_block(rest[:, i].copy())), bits(y[:, i])), i
    step(3, settle=True)
    assert rows[-1] == 1, rows                             # (`damp` keeps the row its schedule needs: the loop re-loads it by itself)
    both("decay", 0.2)
    step()
    assert rows[-1] == 3
    step(3, settle=True)
    assert rows[-1] == 2
    assert b.ood_flags() == 0 and b.tier_note().startswith("translated to gfx950 code")
    for i, o in oracles.items():
        assert b.instruction_counter_i(i) == o.instruction_counter()
    assert b.info("xlate_builds") <= 4, b.info("xlate_builds")     # (the first code; `g`'s row; the schedule's code; the long block's class of block lengths when staged)


def test_damaged_state_images_are_refused_before_any_address_is_computed(gpu):
    """fxb_load_state reads a header the CALLER supplies (a checkpoint file): every field is validated before it enters pointer
    arithmetic - negative or absurd slot counts (which would make the size check accept a short buffer and the section
    pointers run backwards), a wrong version, other register counts, a truncated image.  Each refusal leaves the handle as it
    was: the next block continues bit-exactly.  (negative iSlots / xSlots used to pass both checks.)"""
    import struct
    text = ("itramsize 37 \nxtramsize 100 \nstatic rd\nstatic xd\nstatic a\ninput in 0\noutput out 0\n"
            "idelay read, rd, at, 0\nxdelay read, xd, at, 0\nmacs a, in, rd, 0.5\nidelay write, a, at, 0\nxdelay write, a, at, 0\nmacs out, a, xd, 0.25\nend")
    N, S = 130, 60
    x = progs.stimulus(N, 3 * S)
    b = gpu.Batch(N, 1, 0)
    twin = gpu.Batch(N, 1, 0)
    assert b.load_text(text) and twin.load_text(text)
    b.process_block(x[:S])
    twin.process_block(x[:S])
    img = b.save_state()
    # header: magic u32, version u32, n i64, channels, nRegs, stateRows, iSlots, xSlots i32, reserved[7]
    magic, version, n, channels, n_regs, rows, islots, xslots = struct.unpack_from("<IIqiiiii", img, 0)
    assert (magic, version, n, channels) == (0x54535846, 1, N, 1) and islots == 37 and xslots == 100

    def mutated(**kw):
        f = dict(magic=magic, version=version, n=n, channels=channels, n_regs=n_regs, rows=rows, islots=islots, xslots=xslots)
        f.update(kw)
        out = img.copy()
        struct.pack_into("<IIqiiiii", out, 0, f["magic"], f["version"], f["n"], f["channels"], f["n_regs"], f["rows"], f["islots"], f["xslots"])
        return out

    bad = [mutated(islots=-37), mutated(xslots=-100), mutated(islots=-(rows + 100)), mutated(xslots=-2 ** 31), mutated(islots=2 ** 30), mutated(version=2),
           mutated(magic=0), mutated(n=N + 1), mutated(n=-N), mutated(channels=0), mutated(n_regs=n_regs + 1), mutated(rows=rows + 1), mutated(rows=-1), mutated(rows=0),
           mutated(islots=38), img[:-4], img[:64], img[:10], mutated(islots=-37)[:64 + N * 4 * (rows + 100 - 37)]]
    for k, image in enumerate(bad):
        with pytest.raises(RuntimeError):
            b.load_state(image)
        assert b.last_error(), k
    # the handle is where it was: the next blocks equal an untouched twin's, and the good image still loads
    assert np.array_equal(bits(b.process_block(x[S:2 * S])), bits(twin.process_block(x[S:2 * S])))
    b.load_state(img)
    twin.load_state(img)
    assert np.array_equal(bits(b.process_block(x[2 * S:])), bits(twin.process_block(x[2 * S:])))
    # a smaller delay line in the image than in the batch is legal (the rest stays zero) - and not a way around the size check
    assert b.info("itram_slots") == 37


def test_delay_memory_that_cannot_be_allocated(gpu):
    """a program whose delay lines do not fit the device - xtramsize 1 048 576 x 262 144 instances = 1 TiB on a 288 GB GPU: the
    first block (and fxb_prepare) returns FX_E_MEMORY with the runtime's message, nothing of the attempt stays allocated, the
    handle stays alive and usable - as the reference's object does after a failed loadFile (source/FX8010.cpp:777-875: `false`
    + error list, object intact): a further load that shrinks the line (the loader's size is whatever the last xtramsize line
    said) runs, bit-exactly.  Then 50 create / load / run / destroy cycles leave hipMemGetInfo where it was.  (The same paths
    under AddressSanitizer with an injected allocator, and module-load failures, without a GPU:
    tests/test_host_sanitizers.py.)"""
    import torch
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    N = 262144
    big = ("xtramsize 1048576 \ninput in 0\noutput out 0\nstatic xd\nstatic a\nxdelay read, xd, at, 0\nmacs a, in, xd, 0.5\nxdelay write, a, at, 0\nmacs out, a, 0, 0\nend")
    b = gpu.Batch(N, 1, 0)
    assert b.load_text(big), b.errors()
    free0, total = torch.cuda.mem_get_info(0)
    assert total < N * 1048576 * 4
    x = progs.stimulus(N, 8)
    for attempt in range(2):
        with pytest.raises(RuntimeError) as e:
            b.process_block(x)
        assert "(-5)" in str(e.value) and "TRAM" in str(e.value), str(e.value)     # FX_E_MEMORY, "hipMalloc TRAM: out of memory"
    with pytest.raises(RuntimeError) as e:
        b.prepare(8, True)
    assert "(-5)" in str(e.value)
    free1, _ = torch.cuda.mem_get_info(0)
    assert abs(free1 - free0) < (256 << 20), (free0, free1)                         # nothing of the attempt is left (the state rows are there in both readings)
    assert b.load_text("xtramsize 64 \nend"), b.errors()                            # the handle lives: the accumulated program with a line that fits
    y = b.process_block(x)
    for n in (0, 63, 64, N - 1):
        o = Oracle(1)
        assert o.load_text(big) and o.load_text("xtramsize 64 \nend")
        assert np.array_equal(bits(o.process_block(x[:, n].copy())), bits(y[:, n])), n
    assert b.ood_flags() == 0 and b.info("xtram_slots") <= 64
    del b
    # create / load / run / destroy: the device's free memory comes back to where it was
    def cycle():
        c = gpu.Batch(20000, 1, 0)
        assert c.load_text(progs.CONFIGS["config5"]())
        c.process_block(progs.stimulus(20000, 4))
        c.set_register_i("u", 5, 0.5)
        c.process_block(progs.stimulus(20000, 4))
        c.close()
    cycle()
    torch.cuda.synchronize()
    base, _ = torch.cuda.mem_get_info(0)
    for _ in range(50):
        cycle()
    torch.cuda.synchronize()
    after, _ = torch.cuda.mem_get_info(0)
    assert abs(after - base) <= (8 << 20), (base, after)


@pytest.mark.parametrize("channels", [1, 2])
def test_pinned_caller_buffers_are_processed_in_place(gpu, channels, monkeypatch):
    """A real-time host keeps its PCM in pinned memory; fxb_process_block then runs the kernel on the caller's buffers themselves
    (hipPointerGetAttributes says they are device-visible: no staging copies, one launch - fx_batch.cpp processHost).  Same words
    as through pageable buffers (staged copies) and as with FX_HOST_PIPELINE=0 (the knob that turns the in-place path off):
    uneven blocks from 1 sample up, a slider moving in between, a control schedule inside a block, a buffer that starts in the middle of a pinned allocation, input
    and output in ONE buffer (in place in the caller's sense too), mono and stereo; the mono run against the oracle."""
    import torch
    for k in ("FX_KERNEL", "FX_INST_PER_LANE", "FX_HOST_PIPELINE"):
        monkeypatch.delenv(k, raising=False)
    N, cuts = 20000, [0, 1, 33, 65, 600, 640]
    if channels == 1:
        text = progs.config5()
    else:
        text = ("itramsize 100 \ninput inl 0\ninput inr 1\noutput outl 0\noutput outr 1\ncontrol decay = 0.45\nstatic rd\nstatic a\nstatic b\n"
                "idelay read, rd, at, 0\nmacs a, inl, rd, decay\nmacs b, inr, a, 0.5\nidelay write, b, at, 0\ninterp outl, outl, 0.25, a\nmacs outr, b, inl, 0.125\nend")
    S = cuts[-1]
    x = progs.stimulus(N * channels, S).reshape(S, channels, N)
    pin = torch.empty((S + 3, channels, N), dtype=torch.float32).pin_memory()
    xin = pin.numpy()[3:]                       # (a view that does not start at the allocation's base)
    xin[...] = x
    pout = torch.empty((S, channels, N), dtype=torch.float32).pin_memory()
    def run(env, pinned, alias=False):
        if env:
            monkeypatch.setenv("FX_HOST_PIPELINE", env)
        b = gpu.Batch(N, channels, 0)
        monkeypatch.delenv("FX_HOST_PIPELINE", raising=False)
        assert b.load_text(text), b.errors()
        xin[...] = x                                # (the aliasing run below leaves its outputs in this buffer)
        out = []
        for k, (lo, hi) in enumerate(zip(cuts[:-1], cuts[1:])):
            if k == 2:
                b.set_register("decay", 0.3)
            if k == 3:                                  # a schedule inside the longest block (applied by the generated loop itself)
                b.set_register_track("decay", np.array([0.2, 0.5, 0.1, 0.4], dtype=np.float32), 100)
            if pinned and alias:
                buf = pin.numpy()[3 + lo:3 + hi]
                buf[...] = x[lo:hi]
                out.append(b.process_block(buf, buf).copy())
            elif pinned:
                out.append(b.process_block(xin[lo:hi], pout.numpy()[lo:hi]).copy())
            else:
                out.append(b.process_block(x[lo:hi].copy()))
        return np.concatenate(out, axis=0), b
    y_staged, _ = run(None, False)
    y_place, b = run(None, True)
    y_alias, _ = run(None, True, alias=True)
    y_off, _ = run("0", True)
    assert np.array_equal(bits(y_place), bits(y_staged)) and np.array_equal(bits(y_off), bits(y_staged)) and np.array_equal(bits(y_alias), bits(y_staged))
    assert b.ood_flags() == 0
    if channels == 1:
        for inst in (0, 63, 64, N - 1):
            o = Oracle(1)
            assert o.load_text(text)
            ref = []
            for k, (lo, hi) in enumerate(zip(cuts[:-1], cuts[1:])):
                if k == 2:
                    o.set_register("decay", 0.3)
                if k == 3:
                    for t, v in enumerate((0.2, 0.5, 0.1, 0.4)):
                        o.set_register("decay", v)
                        ref.append(o.process_block(x[lo + 100 * t:(lo + 100 * (t + 1) if t < 3 else hi), 0, inst].copy()))
                    continue
                ref.append(o.process_block(x[lo:hi, 0, inst].copy()))
            assert np.array_equal(bits(np.concatenate(ref)), bits(y_place[:, 0, inst])), inst
            assert b.instruction_counter_i(inst) == o.instruction_counter()


def test_overlapping_and_partly_pinned_buffers_take_the_staged_copies(gpu, monkeypatch):
    """The in-place path is for buffers the kernel can work on sample by sample: one buffer (in == out) or two that do not
    overlap, each WHOLLY inside one pinned mapping.  An output range shifted against the input by a few sample periods would be
    overwritten while other wavefronts still read it, and a buffer only the front of which is registered would fault the GPU at its
    first unpinned page: both take the staged copies (fx_batch.cpp processHost: overlapButNotEqual, deviceVisibleRange), so the
    words are those of pageable buffers (or, where the runtime cannot copy from a half-registered buffer either, the call says so).  The reference's caller owns one float per call (/root/reference/include/FX8010.h:57);
    blocks and their aliasing rules are this library's, written in include/fx8010_amd.h."""
    import torch
    for k in ("FX_KERNEL", "FX_INST_PER_LANE", "FX_HOST_PIPELINE"):
        monkeypatch.delenv(k, raising=False)
    N, S, text = 20000, 48, progs.config5()
    x = progs.stimulus(N, S).reshape(S, 1, N)
    def fresh():
        b = gpu.Batch(N, 1, 0)
        assert b.load_text(text), b.errors()
        return b
    want = fresh().process_block(x.copy()).copy()
    pin = torch.empty((S + 7, 1, N), dtype=torch.float32).pin_memory().numpy()
    for shift_in, shift_out in ((0, 1), (5, 0), (0, 7), (2, 2)):       # output behind the input, in front of it, ..., one buffer
        pin[...] = 0
        pin[shift_in:shift_in + S] = x
        got = fresh().process_block(pin[shift_in:shift_in + S], pin[shift_out:shift_out + S]).copy()
        assert np.array_equal(bits(got), bits(want)), (shift_in, shift_out)
    cudart = torch.cuda.cudart()
    if not hasattr(cudart, "cudaHostRegister"):
        pytest.skip("this torch does not expose hipHostRegister")
    page = 4096
    raw = np.zeros(2 * S * N * 4 + 2 * page, dtype=np.uint8)
    base = (raw.ctypes.data + page - 1) // page * page
    buf = np.frombuffer(raw, dtype=np.float32, count=2 * S * N, offset=base - raw.ctypes.data).reshape(2 * S, 1, N)
    out = np.zeros((S, 1, N), dtype=np.float32)
    registered = (S * N * 4) // page * page                              # the front of the buffer: a little less than S sample periods
    assert int(cudart.cudaHostRegister(base, registered, 0)) == 0
    try:
        for lo in (0, S // 2, S):                                          # ends just past the registration, straddles its end, wholly outside
            buf[...] = 0
            buf[lo:lo + S] = x
            b = fresh()
            try:
                got = b.process_block(buf[lo:lo + S], out).copy()
            except RuntimeError as e:
                # ROCm 7.2's copy engine refuses a source that straddles the end of a registration ("invalid argument"): an error
                # return of this call, where the in-place path would have been a GPU page fault - and the handle carries on
                assert lo <= S // 2 and "invalid argument" in str(e), (lo, str(e))
                got = b.process_block(x.copy()).copy()
            assert np.array_equal(bits(got), bits(want)), lo
            assert b.ood_flags() == 0
        # wholly inside it (and the output pinned as well): in place, the same words
        pout = torch.empty((S // 2, 1, N), dtype=torch.float32).pin_memory().numpy()
        buf[:S] = x
        got = fresh().process_block(buf[:S // 2], pout).copy()
        assert np.array_equal(bits(got), bits(want[:S // 2]))
    finally:
        assert int(cudart.cudaHostUnregister(base)) == 0


def test_host_alloc_gives_buffers_that_are_processed_in_place(gpu):
    """fxb_host_alloc / fxb_host_free: pinned PCM buffers for hosts that do not link the HIP runtime themselves (the reference's
    callers hold their audio in std::vector<float>, include/FX8010.h:57).  Blocks on them equal blocks on pageable arrays bit for
    bit, input and output may be one buffer, a freed buffer's size comes back to the device's host-visible pool (200 cycles)."""
    text = progs.config3()
    N, S = 5000, 70
    x = progs.stimulus(N, 2 * S)
    a = gpu.Batch(N, 1, 0)
    b = gpu.Batch(N, 1, 0)
    assert a.load_text(text) and b.load_text(text)
    bin_, bout = gpu.HostBuffer((S, N)), gpu.HostBuffer((S, N))
    for k in range(2):
        ref = a.process_block(x[k * S:(k + 1) * S].copy())
        bin_.array[...] = x[k * S:(k + 1) * S]
        if k == 0:
            got = b.process_block(bin_.array, bout.array)
        else:
            got = b.process_block(bin_.array, bin_.array)          # in place in the caller's sense too
        assert np.array_equal(bits(ref), bits(got)), k
    assert a.instruction_counter() == b.instruction_counter()
    bin_.close()
    bout.close()
    for _ in range(200):
        h = gpu.HostBuffer((1024, 1024))
        h.array[0, 0] = 1.0
        h.close()
    with pytest.raises(RuntimeError):
        gpu.HostBuffer((0,))
