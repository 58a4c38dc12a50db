#!/usr/bin/env python3
"""Sample the GPU's shader clock and socket power (rocm-smi) while a bench.py configuration runs.

    python3 tools/clock_watch.py config5 [steps]

Prints one JSON object: the bench line's value and the clock / power samples taken while it ran (the
chip does not hold 2.4 GHz under the config5 instruction mix)."""
import json
import re
import subprocess
import sys
import threading
import time


def sample():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    sclk = re.search(r"sclk clock level:.*?\((\d+)Mhz\)", out)
    power = re.search(r"Power \(W\):\s*([0-9.]+)", out)
    return (int(sclk.group(1)) if sclk else None, float(power.group(1)) if power else None, out if not sclk else None)


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "config5"
    steps = sys.argv[2] if len(sys.argv) > 2 else "100"
    idle = sample()
    proc = subprocess.Popen([sys.executable, "bench.py", "--no-extras", "--config", config, "--steps", steps, "--cpu-seconds", "0",
                             "--parity-instances", "0"], stdout=subprocess.PIPE, text=True)
    samples = []
    t0 = time.time()
    while proc.poll() is None:
        s = sample()
        samples.append((round(time.time() - t0, 2), s[0], s[1]))
        time.sleep(0.2)
    line = json.loads(proc.stdout.read().strip().splitlines()[-1])
    print(json.dumps({"config": config, "value": line["value"], "ms_per_step": line["ms_per_step"], "idle": idle[:2], "raw": idle[2],
                      "samples": samples}))


if __name__ == "__main__":
    main()
