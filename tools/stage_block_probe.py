"""Short blocks and barrier spacing of staged programs (GPU box): time per launch of config2 (or another probe program) for block
lengths S and stage counts K, with the library's step length for the block class or a pinned one (FX_STAGES_GROUP).
    python tools/stage_block_probe.py [program] [instances]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fx8010-emulator-core_amd", "python"), os.path.join(ROOT, "oracle")]
import torch  # noqa: E402,F401
import fx8010_amd as A  # noqa: E402
import fx8010_programs as P  # noqa: E402


def run(text, n, S, K, group=None, launches=60):
    os.environ["FX_STAGES"] = str(K)
    if group:
        os.environ["FX_STAGES_GROUP"] = str(group)
    else:
        os.environ.pop("FX_STAGES_GROUP", None)
    b = A.Batch(n, 1, 0)
    assert b.load_text(text)
    x = torch.from_numpy(P.stimulus(n, S)).cuda()
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    ms = []
    for _ in range(launches):
        b.process_block_dev(x.data_ptr(), y.data_ptr(), S)
        b.sync()
        ms.append(b.last_kernel_ms())
    return float(np.median(ms[5:])), b.info("waves_per_wg"), b.info("lds_bytes_per_wg")


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "config2"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    text = (P.CONFIGS.get(name) or P.PROBE_PROGRAMS[name])()
    print("%s, %d instances: kernel us per launch (stages actually used)" % (name, n))
    for S in (8, 16, 32, 64, 128, 256, 1024):
        row = []
        for K in (1, 2, 4, 8):
            ms, k, lds = run(text, n, S, K)
            row.append("K=%d: %7.1f (%d)" % (K, ms * 1e3, k))
        print("S=%5d  " % S + "   ".join(row), flush=True)
    print("barrier spacing at S = 2048 (FX_STAGES_GROUP): us per launch")
    for K in (2, 4, 8):
        row = []
        for g in (1, 2, 4, None):
            ms, k, lds = run(text, n, 2048, K, g, launches=12)
            row.append("group %s: %7.1f" % (g or 8, ms * 1e3))
        print("K=%d  " % K + "   ".join(row), flush=True)


if __name__ == "__main__":
    main()
