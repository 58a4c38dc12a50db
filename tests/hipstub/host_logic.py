#!/usr/bin/env python3
"""The HOST LOGIC of libfx8010_amd.so through its C ABI on a machine without a GPU (TEST INFRASTRUCTURE).

Runs against csrc/build/stub/libfx8010_amd.so: the library's unchanged host sources linked with tests/hipstub/ instead of the HIP
runtime (see hip_stub.cpp).  "Device memory" is host memory there and the fill / reduce helper kernels do their real work, so
everything the host engine decides can be observed: register values and rows, code-cache and builder-thread counters, tiers and
their reasons, state images, shard routing, error codes.  What it cannot say anything about is PCM results - the stand-in kernel
copies its input to its output - parity is the GPU tests' business.  Started by tests/test_host_logic_stub.py with FX8010_AMD_LIB
pointing at the stand-in build; prints one line per scenario and exits non-zero on the first failure.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "fx8010-emulator-core_amd", "python"), os.path.join(ROOT, "oracle")]
import fx8010_amd as A  # noqa: E402
import fx8010_programs as P  # noqa: E402

HDR = "input in 0\noutput out 0\ncontrol vol = 0.5\nstatic a\nstatic b\n"


def scenario_registers():
    """setRegisterValue / getRegisterValue semantics on a batch: broadcast, per instance, whole arrays; the reference's return
    codes (source/FX8010.cpp:236-266: 0 found, 1 not found; getRegisterValue of an unknown key is 1.0)"""
    n = 200
    b = A.Batch(n, 1, 0)
    assert b.load_text(HDR + "control mix = 0.25\nmacs a, a, vol, in\ninterp b, b, vol, a\nmacs out, b, a, mix\nend"), b.errors()
    assert b.get_register_i("vol", 0) == np.float32(0.5) and b.get_register_i("nosuch", 0) == 1.0
    assert b.set_register("vol", 0.75) == 0 and b.set_register("nosuch", 0.1) == 1
    assert b.get_register_i("vol", 199) == np.float32(0.75)
    assert b.set_register_i("vol", 63, 0.125) == 0
    assert b.get_register_i("vol", 63) == np.float32(0.125) and b.get_register_i("vol", 64) == np.float32(0.75)
    vals = np.linspace(0, 1, n).astype(np.float32)
    assert b.set_register_array("mix", vals) == 0
    assert np.array_equal(b.get_register_array("mix"), vals)
    x = P.stimulus(n, 16)
    b.process_block(x)
    assert b.get_register_i("vol", 63) == np.float32(0.125)      # the row survives the first lowering
    assert b.set_register("vol", 0.3) == 0 and b.get_register_i("vol", 63) == np.float32(0.3)   # a broadcast write levels it again
    for bad in (-1, n):
        try:
            b.set_register_i("vol", bad, 0.1)
            raise AssertionError("instance %d accepted" % bad)
        except RuntimeError as e:
            assert "(-3)" in str(e)
    assert b.ood_flags() == 0 and b.instruction_counter() == 0   # (no real kernel ran)


def scenario_controls_and_code_cache(builder):
    """moving controls become rows once, ahead of time on the builder thread (first touch = a pointer swap); without the
    thread one re-translation; a control no instruction reads never gets a row; a shape that comes back is a swap"""
    if not builder:
        os.environ["FX_BUILDER"] = "0"
    try:
        b = A.Batch(70, 1, 0)
    finally:
        os.environ.pop("FX_BUILDER", None)
    assert b.load_text(HDR + "control mix = 0.25\ncontrol unused = 0.5\nmacs a, a, vol, in\ninterp b, b, vol, a\nmacs out, b, a, mix\nend"), b.errors()
    x = P.stimulus(70, 8)
    builds, rows, tiers = [], [], []
    for blk in range(30):
        if 3 <= blk < 25:
            b.set_register("vol", 0.1 + 0.03 * blk)
        if blk in (10, 11, 20):
            b.set_register("mix", 0.5 - 0.01 * blk)
            b.set_register("unused", 0.01 * blk)
        if blk == 2 and builder:
            b.prepare(8, True)                         # (the builder's follow-up is finished: the swap below is deterministic)
        b.process_block(x)
        builds.append(b.info("xlate_builds"))
        rows.append(b.info("num_rows"))
        tiers.append(b.info("kernel"))
    assert all(t >= 9 for t in tiers), tiers
    if builder:
        assert builds[-1] == 1 and b.info("xlate_background_builds") >= 1 and b.info("code_cache_hits") >= 1, (builds, b.info("xlate_background_builds"))
    else:
        assert builds[2] == 1 and builds[3] == 2 and builds[-1] == 2 and b.info("xlate_background_builds") == 0, builds
    assert rows[3] == rows[2] + 2 and rows[-1] == rows[3], rows          # vol and mix; `unused` never gets one
    assert b.get_register_i("unused", 7) == np.float32(0.2)


def scenario_lean_control_variant():
    """control mode puts every declared control in a row at the first touch of one (a pointer swap to code built ahead); the
    controls that have NOT moved lately go back into the code a few blocks later (the lean variant, from the builder thread, another
    swap); a second control that starts moving, a per-instance write or a state image brings the full variant back at once -
    never a translation on the caller's thread (FXB_INFO_CONTROL_ROWS says which variant is in force)"""
    n = 262144 // 64      # (the plain program, no stages)
    b = A.Batch(n, 1, 0)
    assert b.load_text(P.config5()), b.errors()           # controls damp, decay, diff
    x = P.stimulus(n, 32)
    b.prepare(32, True)
    b.process_block(x)
    assert b.info("control_rows") == 0 and b.info("xlate_builds") == 1
    def settle(want):
        for _ in range(3):
            b.prepare(32, True)                            # (the builder's queue is empty)
            b.process_block(x)
        assert b.info("control_rows") == want, (b.info("control_rows"), want)
    b.set_register("decay", 0.4)
    b.process_block(x)
    assert b.info("control_rows") == 3                    # the whole panel, at once
    settle(1)                                              # ... then only the one that moves
    hits = b.info("code_cache_hits")
    for k in range(20):                                    # the slider keeps moving: fills of its row, no change of code
        b.set_register("decay", 0.1 + 0.01 * k)
        b.process_block(x)
    assert b.info("control_rows") == 1 and b.info("code_cache_hits") == hits
    held = b.info("num_rows")
    b.set_register_i("w0", 3, 0.1)                        # (a register the program writes anyway: nothing changes)
    b.process_block(x)
    assert b.info("num_rows") == held and b.info("control_rows") == 1
    b.set_register("damp", 0.2)                            # a second control starts moving: the full variant is still there
    b.process_block(x)
    assert b.info("control_rows") == 3
    settle(2)
    b.set_register_i("diff", 5, 0.55)                      # the cold one gets per-instance values: it needs its row NOW
    b.process_block(x)
    assert b.info("control_rows") == 3
    settle(3)
    assert b.get_register_i("diff", 5) == np.float32(0.55) and b.get_register_i("diff", 6) == np.float32(0.6)
    assert b.info("xlate_builds") == 1, b.info("xlate_builds")   # every other build came from the builder thread
    # controls that are left alone cool down (8192 sample periods): their values go back into the code as well - `diff` keeps
    # its row while its instances hold different values, and loses it after a broadcast write has levelled them and time has passed
    for _ in range(8192 // 32 + 40):
        b.process_block(x)
    settle(1)
    b.set_register("diff", 0.6)
    b.process_block(x)
    assert b.info("control_rows") in (1, 3)               # (the row was there: a fill)
    for _ in range(8192 // 32 + 40):
        b.process_block(x)
    settle(0)
    b.set_register("decay", 0.33)                          # ... and the first touch after a rest is the full variant again, at once
    b.process_block(x)
    assert b.info("control_rows") == 3
    assert b.info("xlate_builds") == 1, b.info("xlate_builds")
    # a control that moved again after it had cooled down rests twice as long before it is folded in the next time (a slider
    # that moves every few hundred milliseconds must not have code built for it again and again beside a real-time stream)
    for _ in range(8192 // 32 + 40):
        b.process_block(x)
    settle(1)                                              # 8192 sample periods later: `decay` still has its row
    for _ in range(8192 // 32 + 40):
        b.process_block(x)
    settle(0)                                              # ... 16384: folded in
    assert b.info("xlate_builds") == 1, b.info("xlate_builds")
    # a state image into a fresh handle in control mode: whatever was lean is full again until the builder has caught up
    c = A.Batch(n, 1, 0)
    assert c.load_text(P.config5())
    c.process_block(x)
    c.set_register("decay", 0.4)
    c.process_block(x)
    for _ in range(3):
        c.prepare(32, True)
        c.process_block(x)
    assert c.info("control_rows") == 1
    b.set_register_i("diff", 5, 0.55)
    b.process_block(x)
    img = b.save_state()
    c.load_state(img)
    c.process_block(x)
    assert c.info("control_rows") == 3 and c.get_register_i("diff", 5) == np.float32(0.55)


def scenario_block_classes():
    """code is generated for a class of block lengths; a host that alternates between two classes translates each once"""
    b = A.Batch(300, 1, 0)
    assert b.load_text(P.config2())
    for rep in range(6):
        for s in (16, 400):
            x = P.stimulus(300, s)
            for _ in range(5):
                b.process_block(x)
    # (the builder thread also makes the stage counts on trial and the control-row variants of each: they never hold a block up)
    assert b.info("xlate_builds") <= 2 and b.info("xlate_background_builds") >= 1, (b.info("xlate_builds"), b.info("xlate_background_builds"))
    assert b.info("code_cache_hits") >= 8 and b.info("code_cached") <= 9, (b.info("code_cache_hits"), b.info("code_cached"))
    assert b.info("waves_per_wg") >= 2 and "stages" in b.tier_note()


def scenario_tiers():
    """fxb_tier_note / FXB_INFO_KERNEL: which tier runs a program and why (no device needed to decide)"""
    x = P.stimulus(70, 8)
    b = A.Batch(70, 1, 0)
    assert b.tier_note() == "no program loaded"
    assert b.load_text(P.config3())
    assert "not lowered yet" in b.tier_note()
    b.process_block(x)
    assert b.info("kernel") >= 9 and b.tier_note().startswith("translated to gfx950 code (fx_xlate_v"), b.tier_note()
    m = A.Batch(70, 1, 0)
    assert m.load_text("input in 0\noutput out 0\nstatic a\nmacs a, in, 0, 0\nskip ccr, ccr, 6, 2\nmacs out, a, 0, 0\nend")   # the SKIP can jump over END
    m.process_block(x)
    assert 2 <= m.info("kernel") <= 8 and "END can be skipped" in m.tier_note(), m.tier_note()
    g = A.Batch(70, 1, 0)
    assert g.load_text("input in 0\noutput out 0\nstatic a\nlog a, in, 40, 0\nmacs out, a, 0, 0\nend")
    g.process_block(x)
    assert g.info("kernel") == 0 and "out-of-range table" in g.tier_note(), g.tier_note()
    for knob, lo, hi in (("asm", 2, 8), ("asm_lds", 1, 1), ("hip", 0, 0), ("xlate_v168", 14, 14)):
        os.environ["FX_KERNEL"] = knob
        try:
            k = A.Batch(70, 1, 0)
        finally:
            os.environ.pop("FX_KERNEL")
        assert k.load_text(P.config2())
        k.process_block(x)
        assert lo <= k.info("kernel") <= hi, (knob, k.info("kernel"), k.tier_note())
    big = A.Batch(70, 1, 0)     # 300 per-instance registers: beyond the largest VGPR build -> the LDS interpreter
    text = "input in 0\noutput out 0\n" + "".join("static r%d\n" % i for i in range(300)) + "".join("macs r%d, in, r%d, 0.5\n" % (i, i) for i in range(300)) + "macs out, r0, r299, 0.5\nend"
    assert big.load_text(text)
    big.process_block(x)
    assert big.info("kernel") == 1 and "register file in LDS" in big.tier_note(), big.tier_note()


def scenario_state_images():
    """fxb_save_state / fxb_load_state: the register part of an image round-trips (per-instance and broadcast values), an image
    of a three-shard handle loads into a single one and back"""
    n = 300
    text = "itramsize 37 \n" + HDR + "static rd\nidelay read, rd, at, 0\nmacs a, in, rd, vol\nidelay write, a, at, 0\nmacs out, a, b, 0.5\nend"
    one = A.Batch(n, 1, 0)
    three = A.Batch(n, 1, devices=[0, 0, 0])
    assert one.load_text(text) and three.load_text(text)
    x = P.stimulus(n, 8)
    for h in (one, three):
        h.process_block(x)
        h.set_register("vol", 0.25)
        h.set_register_i("b", 129, 0.5)
        h.seed_noise_i(5, 111, 222)
    img1, img3 = one.save_state(), three.save_state()
    assert img1.size == img3.size and np.array_equal(img1, img3)           # laid out by GLOBAL instance: the partition does not show
    one.set_register("vol", 0.9)
    one.set_register_i("b", 129, -0.5)
    one.load_state(img3)
    assert one.get_register_i("vol", 0) == np.float32(0.25) and one.get_register_i("b", 129) == np.float32(0.5) and one.get_register_i("b", 130) == 0.0
    three.load_state(img1)
    assert three.get_register_i("b", 129) == np.float32(0.5) and three.get_cursors_i(299) == one.get_cursors_i(299)
    assert np.array_equal(three.get_tram_i(0, 200, 37), np.zeros(37, dtype=np.float32))
    small = A.Batch(n - 1, 1, 0)
    assert small.load_text(text)
    small.process_block(P.stimulus(n - 1, 8))
    try:
        small.load_state(img1)
        raise AssertionError("an image of another instance count was accepted")
    except RuntimeError as e:
        assert "(-3)" in str(e)


def scenario_shards():
    """a multi-shard handle routes per-instance calls to the owning shard; the partition is the library's (whole wavefronts)"""
    n = 1000
    b = A.Batch(n, 1, devices=[0, 0, 0, 0])
    assert [c for _, _, c in b.shards()] == [256, 256, 256, 232] and [f for _, f, _ in b.shards()] == [0, 256, 512, 768]
    assert b.load_text(HDR + "macs a, a, vol, in\nmacs out, a, b, 0.5\nend")
    for inst in (0, 255, 256, 767, 768, 999):
        assert b.set_register_i("b", inst, inst / 1000.0) == 0
    for inst in (0, 255, 256, 767, 768, 999):
        assert b.get_register_i("b", inst) == np.float32(inst / 1000.0)
    assert b.get_register_i("b", 1) == 0.0
    vals = np.arange(n, dtype=np.float32) / 2048
    b.set_register_array("a", vals)
    assert np.array_equal(b.get_register_array("a"), vals)
    x = P.stimulus(n, 8)
    y = b.process_block(x)
    assert np.array_equal(x, y)                                           # every shard's columns went in and came back
    assert A.shard_plan(2097152, 8) == [(i * 262144, 262144) for i in range(8)]


def scenario_errors():
    """error codes of the boundary: nothing loaded, bad arguments, a program the lowering refuses, a failed load's error list
    (the reference: loadFile returns false and the list says why, source/FX8010.cpp:777-875)"""
    b = A.Batch(64, 1, 0)
    x = P.stimulus(64, 4)
    try:
        b.process_block(x)
        raise AssertionError("a block without a program was accepted")
    except RuntimeError as e:
        assert "(-2)" in str(e)                                           # FX_E_NOTREADY
    assert not b.load_text("input in 0\noutput out 0\nmacs out, nosuch, 0, 0\nend")
    errs = b.errors()
    assert errs[0][0] == "Kein Fehler" and any("nicht deklariert" in d for d, _ in errs[1:]), errs
    assert b.set_register("in", 0.25) == 0 and b.get_register_i("in", 3) == np.float32(0.25)   # the object lives; what the failed load declared stays declared
    try:
        A.Batch(0, 1, 0)
        raise AssertionError("an empty batch was created")
    except RuntimeError:
        pass
    big = A.Batch(64, 1, 0)
    assert big.load_text("itramsize 9000 \ninput in 0\noutput out 0\nstatic rd\nidelay read, rd, at, 0\nmacs out, in, rd, 0.5\nend")   # the reference's size check is ineffective: accepted ...
    try:
        big.process_block(x)
        raise AssertionError("a delay line beyond smallDelayBuffer[8192] was lowered")
    except RuntimeError as e:
        assert "(-4)" in str(e)                                           # ... and refused by the lowering (FX_E_PROGRAM)


def main():
    lib = A.load()
    assert "stub" in os.path.abspath(A.LIB_PATH), "run with FX8010_AMD_LIB = the stand-in build (csrc/build/stub)"
    scenarios = [("registers", scenario_registers), ("controls, builder thread", lambda: scenario_controls_and_code_cache(True)),
                 ("controls, no builder", lambda: scenario_controls_and_code_cache(False)), ("lean control variant", scenario_lean_control_variant),
                 ("block classes", scenario_block_classes),
                 ("tiers", scenario_tiers), ("state images", scenario_state_images), ("shards", scenario_shards), ("errors", scenario_errors)]
    wanted = sys.argv[1:]
    for name, fn in scenarios:
        if wanted and not any(w in name for w in wanted):
            continue
        fn()
        print("%-26s ok" % name, flush=True)
    print("host logic: all scenarios passed (%d HIP devices in the stand-in)" % lib.fxb_device_count())


if __name__ == "__main__":
    main()
