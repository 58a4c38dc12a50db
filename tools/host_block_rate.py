"""Rate of fxb_process_block on HOST buffers (PCIe included): one large block of a benchmark program, with the copies of its
pieces overlapping the kernel (default) and with FX_HOST_PIPELINE=0 (copy in, kernel, copy out).  DESIGN.md section 5.

    python tools/host_block_rate.py [config] [instances] [samples]
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "fx8010-emulator-core_amd/python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402


def one(config, n, s):
    if os.environ.get("FX_HOST_RATE_PINNED") == "1":
        import torch  # noqa: F401  (before the library initialises HIP)
    import fx8010_amd as A
    import fx8010_programs as P
    from pyoracle import Oracle
    text = P.CONFIGS[config]()
    x = P.stimulus(n, s)
    out = np.zeros_like(x)   # (touched: a fresh np.empty would add the page faults of 65 536 pages to every call)
    if os.environ.get("FX_HOST_RATE_PINNED") == "1":   # caller buffers in pinned memory (torch): copies at DMA rate
        import torch
        xin = torch.empty(x.shape, dtype=torch.float32).pin_memory()
        xin.numpy()[...] = x
        x = xin.numpy()
        keep = torch.empty(x.shape, dtype=torch.float32).pin_memory()
        out = keep.numpy()
    b = A.Batch(n, 1, 0)
    assert b.load_text(text)
    y = b.process_block(x, out)      # first call: allocation, translation
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        y = b.process_block(x, out)
    dt = (time.perf_counter() - t0) / reps
    # parity of the last call for a few instances (state carried over the four calls)
    ok = True
    for inst in (0, n // 2, n - 1):
        o = Oracle(1)
        o.load_text(text)
        for _ in range(reps):
            o.process_block(x[:, inst].copy())
        ref = o.process_block(x[:, inst].copy())
        ok = ok and np.array_equal(ref.view(np.uint32), y[:, inst].view(np.uint32))
    instr = b.info("num_instructions")
    return {"pipeline": os.environ.get("FX_HOST_PIPELINE", "1"), "caller_buffers": "pinned" if os.environ.get("FX_HOST_RATE_PINNED") == "1" else "pageable", "ms_per_block": round(dt * 1e3, 2), "host_GBps_each_way": round(x.nbytes / dt / 1e9, 2),
            "mips": round(instr * n * s / dt / 1e6, 1), "kernel_ms_last_piece": round(b.last_kernel_ms(), 3), "parity_ok": bool(ok)}


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "config3"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    s = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    if os.environ.get("FX_HOST_RATE_CHILD"):
        print(json.dumps(one(config, n, s)))
        return
    out = {"config": config, "instances": n, "samples": s, "block_MB_each_way": round(n * s * 4 / 1e6, 1), "runs": []}
    for mode, pinned in (("1", "0"), ("0", "0"), ("1", "1"), ("0", "1")):
        env = dict(os.environ, FX_HOST_PIPELINE=mode, FX_HOST_RATE_PINNED=pinned, FX_HOST_RATE_CHILD="1")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), config, str(n), str(s)], env=env, capture_output=True, text=True)
        out["runs"].append(json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": r.stderr[-400:]})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
