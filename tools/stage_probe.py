"""Stage pipelining probe (GPU box): a benchmark program at its BASELINE size with FX_STAGES = 1, 2, 4, ..., checked against the
oracle on a few instances, kernel time per launch.   python tools/stage_probe.py [config2] [instances] [samples] [stages ... (0 = the library's policy)]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fx8010-emulator-core_amd", "python"), os.path.join(ROOT, "oracle")]
try:
    import torch  # noqa: F401  (torch first: see tests/conftest.py)
except ImportError:
    pass
import fx8010_amd as A  # noqa: E402
import fx8010_programs as P  # noqa: E402
from pyoracle import Oracle  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "config2"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    stages = [int(v) for v in sys.argv[4:]] or [1, 2, 4, 8, 16]
    text = (P.CONFIGS.get(name) or P.PROBE_PROGRAMS[name])()
    x = P.stimulus(n, S)
    check = sorted(set([0, 1, 63, 64, n // 2, n - 1]))
    refs = {}
    for i in check:
        o = Oracle(1)
        assert o.load_text(text)
        r1 = o.process_block(x[:, i].copy())
        r2 = o.process_block(x[:, i].copy())
        refs[i] = (r1, r2, o.instruction_counter())
    for K in stages:
        if K > 0:
            os.environ["FX_STAGES"] = str(K)
        else:
            os.environ.pop("FX_STAGES", None)   # 0: the library's own policy (instances, block length)
        b = A.Batch(n, 1, 0)
        assert b.load_text(text), b.errors()
        y1 = b.process_block(x)
        y2 = b.process_block(x)
        ms = []
        import torch
        xd = torch.from_numpy(x).cuda()
        yd = torch.empty_like(xd)
        torch.cuda.synchronize()
        if K == 0:   # the library's own choice: let it time the options its cost model cannot tell apart (Batch::noteLaunchTime)
            for _ in range(14):
                b.process_block_dev(xd.data_ptr(), yd.data_ptr(), S)
                b.sync()
        for _ in range(6):   # device-resident PCM: ONE launch of S samples, as bench.py times it
            b.process_block_dev(xd.data_ptr(), yd.data_ptr(), S)
            b.sync()
            ms.append(b.last_kernel_ms())
        ms = ms[1:]
        ok = all(np.array_equal(refs[i][0].view(np.uint32), y1[:, i].view(np.uint32)) and np.array_equal(refs[i][1].view(np.uint32), y2[:, i].view(np.uint32))
                 for i in check)
        instr = b.info("num_instructions")
        print("%s n=%d S=%d FX_STAGES=%d -> waves/wg %d kernel %d lds %d: %.3f ms (min %.3f) = %.3f e12 instr/s  parity %s ood %d%s" % (
            name, n, S, K, b.info("waves_per_wg"), b.info("kernel"), b.info("lds_bytes_per_wg"), float(np.median(ms)), min(ms),
            instr * S * n / (min(ms) * 1e-3) / 1e12, ok, b.ood_flags(), "  (%d trial launches)" % b.info("stage_trials") if K == 0 else ""), flush=True)


if __name__ == "__main__":
    main()
