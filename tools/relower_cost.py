"""Cost of a control change (fxb_set_register of a uniform register = re-lowering) per kernel tier.

    python tools/relower_cost.py
"""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fx8010-emulator-core_amd/python"))
import numpy as np  # noqa: E402

import fx8010_amd as A  # noqa: E402
import fx8010_programs as P  # noqa: E402

for tier in ("default", "xlate", "asm", "hip"):
    os.environ.pop("FX_KERNEL", None)
    if tier != "default":
        os.environ["FX_KERNEL"] = tier  # "xlate" pins the translator even while controls move
    for name, N, S in (("config1_shipped", 1, 8), ("config5", 4096, 8)):
        text = P.CONFIGS[name]()
        b = A.Batch(N, 1, 0)
        assert b.load_text(text)
        ctl = b.controls()[0] if b.controls() else None
        x = P.stimulus(N, S)
        b.process_block(x)
        t0 = time.perf_counter()
        for i in range(50):
            b.process_block(x)
        t_plain = (time.perf_counter() - t0) / 50
        t0 = time.perf_counter()
        for i in range(50):
            if ctl:
                b.set_register(ctl, 0.1 + 0.01 * i)
            b.process_block(x)
        t_ctl = (time.perf_counter() - t0) / 50
        print("%-7s %-16s kernel %2d  block %.3f ms   with a control change before it %.3f ms" % (tier, name, b.info("kernel"), t_plain * 1e3, t_ctl * 1e3), flush=True)
        del b
