"""Programs pipelined over the wavefronts of a workgroup (fx_xlate.hpp StageInfo): where the planner cuts, what it refuses,
and the generated code of every stage - re-assembled with llvm-mc, linted for hazards, and checked for the one property a
wrong build would pay for with a hung GPU: every wavefront of the workgroup executes the SAME number of barriers.  CPU only."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gfx950_lint as L  # noqa: E402

import fx8010_amd as A  # noqa: E402
import fx8010_programs as P  # noqa: E402
from test_xlate import assemble, needs_llvm  # noqa: E402

HDR = "input in 0\noutput out 0\ncontrol k = 0.25\nstatic t\nstatic u\n"


def chain(cells, extra=""):
    """a feed-forward chain of one-pole cells, each with a state register of its own: the shape that can be cut"""
    text = HDR + "".join("static s%d\n" % i for i in range(cells)) + extra
    prev = "in"
    for i in range(cells):
        text += "interp s%d, s%d, k, %s\nmacs t, s%d, in, 0.05\n" % (i, i, prev, i)
        prev = "t"
    return text + "macs out, 0, t, 1.0\nend"


def plan(text, stages, vgprs=128, options=0):
    fe = A.FrontEnd(1)
    if options:
        fe.set_option(options)
    assert fe.load_text(text), fe.errors()
    _, _, k, info = fe.translate_staged(stages, 0, 0, vgprs)
    return fe, k, info, fe.last_error()


def test_planner_cuts_feed_forward_programs_only():
    fe, k, info, why = plan(P.CONFIGS["config2"](), 8)
    assert k == 8 and len(info) == 2 * 7 + 1
    cuts, live = info[0:-1:2], info[1:-1:2]
    assert cuts == sorted(cuts) and all(1 <= n <= 3 for n in live)      # `in` and one of t / s_i cross every cut
    # config3 / config4: every state register is used several times per sample (s0 .. s29 four times over): its first read
    # of sample t needs its last write of sample t-1, so no cut between them is legal - the whole program is one recurrence
    for name in ("config3", "config4", "config5"):
        fe, k, info, why = plan(P.CONFIGS[name](), 4)
        assert k == 1 and "no legal cut" in why, (name, why)
    # a backward dependence: u is read at the top and written at the bottom
    fe, k, info, why = plan(HDR + "static a\nstatic b\nmacs a, in, u, 0.5\nmacs b, a, in, 0.25\nmacs t, b, a, 0.5\nmacs t, t, in, 0.1\nmacs t, t, a, 0.1\n"
                            "macs t, t, b, 0.1\nmacs u, t, in, 0.5\nmacs out, 0, u, 1.0\nend", 2)
    assert k == 1
    # the same without the feedback: cut
    fe, k, info, why = plan(HDR + "static a\nstatic b\nmacs a, in, 0.5, 0.5\nmacs b, a, in, 0.25\nmacs t, b, a, 0.5\nmacs t, t, in, 0.1\nmacs t, t, a, 0.1\n"
                            "macs t, t, b, 0.1\nmacs u, t, in, 0.5\nmacs u, u, t, 0.5\nmacs u, u, a, 0.5\nmacs u, u, b, 0.5\nmacs out, 0, u, 1.0\nend", 2)
    assert k == 2
    # delay lines and noise stay in stage 0: a program that ends with its delay-line write cannot be cut
    fe, k, info, why = plan("itramsize 50 \n" + chain(12, "static rd\nidelay read, rd, at, 0\n").replace("macs out, 0, t, 1.0", "idelay write, t, at, 0\nmacs out, 0, rd, 1.0"), 4)
    assert k == 1
    # ... one that starts with read and write can, behind them
    fe, k, info, why = plan("itramsize 50 \n" + HDR + "static rd\nstatic a\n" + "".join("static s%d\n" % i for i in range(12))
                            + "idelay read, rd, at, 0\nmacs a, in, rd, 0.5\nidelay write, a, at, 0\n"
                            + "".join("interp s%d, s%d, k, a\nmacs t, s%d, a, 0.05\n" % (i, i, i) for i in range(12)) + "macs out, 0, t, 1.0\nend", 4)
    assert k >= 2 and info[0] >= 3
    # a SKIP and its shadow are one piece
    text = chain(8).replace("macs t, s3, in, 0.05\n", "macs t, s3, in, 0.05\nskip ccr, ccr, 6, 2\nmacs t, t, 0.5, 0.5\nmacs t, t, in, 0.5\n")
    fe, k, info, why = plan(text, 8)
    assert k >= 2
    fe2 = A.FrontEnd(1)
    assert fe2.load_text(text)
    fe2.lower()
    # (no cut lands between the SKIP and the end of its shadow: the stage boundaries are all outside records of the shadow)


@needs_llvm
@pytest.mark.parametrize("stages", [2, 3, 4, 8])
def test_staged_code_reassembles_lints_and_balances_its_barriers(stages):
    for name, text in (("config2", P.CONFIGS["config2"]()), ("chain_skip", chain(10).replace("macs t, s4, in, 0.05\n", "macs t, s4, in, 0.05\nskip ccr, ccr, 6, 1\nmacs t, t, 0.5, 0.5\n")),
                       ("chain_log", chain(9).replace("macs t, s2, in, 0.05\n", "macs t, s2, in, 0.05\nlog u, t, 3, 0\nexp t, u, 5, 0\n"))):
        fe = A.FrontEnd(1)
        assert fe.load_text(text), name
        res = L.staged_image_listing(fe, stages, 128)
        assert res is not None, name
        listing, size, k, info, heads = res
        for st in range(k):
            for stream in range(5):
                code, lst, _, _ = fe.translate_staged(stages, st, stream, 128)
                if code:
                    assert assemble(lst) == code, (name, st, stream)
        ins = L.disassemble_listing(listing)
        assert ins[-1].addr + ins[-1].size == size
        findings = L.lint_hazards(ins, assume_entry_defs={"vcc", "s62", "s63", "s64", "s65", "s66", "s67"})
        assert not findings, (name, findings[:5])
        problems = L.lint_index_mode(ins, entries=L.stream_entries(ins), any_base=True)
        assert not problems, (name, problems[:5])
        depth = 3
        for st in range(k):
            hs = set(heads[st])
            for h in heads[st]:
                at_head, at_exit = L.barrier_counts(ins, h, hs)
                # a sample ends with the group's barrier or without one (every fourth does: the same samples in every wavefront,
                # decided by the sample counter alone) - never two, never a barrier on one path and not on its sibling
                assert (at_head or at_exit) and at_head <= {0, 1}, (name, st, h, at_head)
                assert at_exit <= {depth * (k - 1 - st), 1 + depth * (k - 1 - st)}, (name, st, h, at_exit)  # the last sample, then the later stages' steps
            assert any(L.barrier_counts(ins, h, hs)[1] for h in heads[st]), (name, st)
        # the cold entries: 3 * stage barriers before the first sample (found as the stubs that end in a branch to a head)
        # -> covered on the GPU by the parity tests; here: the stub of stage k holds exactly 3k barriers
        for st in range(k):
            for stream in range(4):
                code, lst, _, _ = fe.translate_staged(stages, st, stream, 128)
                lines = lst.strip().split("\n")
                stub = []
                for l in reversed(lines[:-1]):
                    if l.startswith(("s_branch", "s_setpc")):
                        break
                    stub.append(l)
                assert sum(1 for l in stub if l == "s_barrier") == depth * st, (name, st, stream)


def test_unstaged_code_is_unchanged_by_the_option():
    fe = A.FrontEnd(1)
    assert fe.load_text(P.CONFIGS["config5"]())
    plain, _ = fe.translate(0, 0)
    again, _, k, _ = fe.translate_staged(4, 0, 0, 0)
    assert k == 1 and again == plain
    assert "s_barrier" not in fe.translate(0, 0)[1]


@needs_llvm
def test_random_feed_forward_programs_plan_verify_lint_and_balance():
    """the corpus of tools/fuzz_stages.py (sections with private state, SKIPs, LOG / EXP, delay lines and noise in the first
    section): every plan that is made passes its symbolic check (a failing one is refused with a 'plan check' message - none
    may occur), every generated stage re-assembles, lints clean and keeps its barriers balanced"""
    import fuzz_stages
    cut = 0
    for seed in range(0, 120):
        rng = np.random.default_rng(seed)
        text = fuzz_stages.random_program(rng, int(rng.integers(2, 9)), int(rng.integers(2, 9)))
        K = int(rng.choice([2, 3, 4, 8]))
        fe = A.FrontEnd(1)
        assert fe.load_text(text), (seed, fe.errors())
        _, _, k, info = fe.translate_staged(K, 0, 0, 128)
        if k < 2:
            assert "plan check" not in fe.last_error(), (seed, fe.last_error())
            continue
        cut += 1
        if seed % 4:
            continue   # (the full check on a quarter of them: it assembles every stream)
        listing, size, k, info, heads = L.staged_image_listing(fe, K, 128)
        ins = L.disassemble_listing(listing)
        assert ins[-1].addr + ins[-1].size == size, seed
        assert not L.lint_hazards(ins, assume_entry_defs={"vcc", "s62", "s63", "s64", "s65", "s66", "s67"}), seed
        assert not L.lint_index_mode(ins, entries=L.stream_entries(ins), any_base=True), seed
        for st in range(k):
            hs = set(heads[st])
            for h in heads[st]:
                at_head, at_exit = L.barrier_counts(ins, h, hs)
                assert at_head <= {0, 1} and at_exit <= {3 * (k - 1 - st), 1 + 3 * (k - 1 - st)}, (seed, st)
    assert cut > 40
