"""What a launch costs beside its samples (GPU box): the kernel's own time (GPU events) against the block length S, device-resident
PCM, fitted as T(S) = a + b * S.  b is the steady rate (what bench.py measures at S = 4096), a is what every launch pays once:
launch, the instance state in and out of HBM (2 x 4 B x (rows + 9) per instance), cold instruction fetch of a loop body of tens
of KB, the first delay-line reads.  A real-time caller's 32-sample block pays a every 666.667 us.
    python tools/block_length_cost.py [config] [instances ...]"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "fx8010-emulator-core_amd", "python")]
import numpy as np
import torch
import fx8010_amd as A
import fx8010_programs as P

name = sys.argv[1] if len(sys.argv) > 1 else "config5"
counts = [int(v) for v in sys.argv[2:]] or [65536, 262144, 393216]
text = P.CONFIGS[name]()
instr = P.count_instructions(text)
for n in counts:
    b = A.Batch(n, 1, 0)
    assert b.load_text(text), b.errors()
    lengths = (1, 2, 4, 8, 16, 32, 64, 128, 256, 1024)
    x = torch.from_numpy(P.stimulus(n, max(lengths))).cuda()
    y = torch.empty_like(x)
    rows = []
    for S in lengths:
        b.prepare(S, False)
        for _ in range(12):
            b.process_block_dev(x.data_ptr(), y.data_ptr(), S)
            b.sync()
        t = []
        for _ in range(60 if S <= 256 else 12):
            b.process_block_dev(x.data_ptr(), y.data_ptr(), S)
            b.sync()
            t.append(b.last_kernel_ms() * 1e3)
        rows.append((S, float(np.median(t)), float(np.min(t))))
    S_, T_ = np.array([r[0] for r in rows], dtype=np.float64), np.array([r[1] for r in rows])
    big = S_ >= 64
    slope, icpt = np.polyfit(S_[big], T_[big], 1)
    print("%s, %d instances (%s; %d register rows)" % (name, n, b.tier_note()[:60], b.info("num_rows")))
    for S, med, mn in rows:
        print("   S = %4d: kernel %9.1f us (fastest %9.1f)   beside its samples at the steady rate: %7.1f us   %5.2f * 10^12 instr/s" % (
            S, med, mn, med - slope * S, instr * S * n / med / 1e6))
    print("   fit over S >= 64: T = %.1f us + %.3f us * S (steady %.2f * 10^12 instr/s); state in + out = %.1f MB" % (
        icpt, slope, instr * n / slope / 1e6, 2 * 4 * (b.info("num_rows") + 9) * n / 1e6))
    b.close()
    del x, y
    torch.cuda.empty_cache()
