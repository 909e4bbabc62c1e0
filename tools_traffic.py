#!/usr/bin/env python3
"""profiles/r2_traffic.json from the memory-side PMC passes of tools_pmc_mem.sh <tag>:

    python tools_traffic.py <tag>

HBM bytes per launch of the dominant kernel = 2 x FETCH_SIZE (gfx950 counts a 128-byte read request as 64,
MI355X_MICROARCH.md, HBM section) + WRITE_SIZE, both in KB per dispatch, averaged over the run's 60 dispatches.
The record names the library it was measured with (sha256 of libgl_hip.so) and the git head: bench.py stamps the
figure only into runs that loaded that very library.
"""
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
tag = sys.argv[1]


def counter(path, name):
    for line in open(path):
        m = re.search(r"regring.*\b%s per-dispatch ([0-9.e+]+)" % name, line)
        if m:
            return float(m.group(1))
    raise SystemExit(f"{name} not found in {path}")


fetch_kb = counter(os.path.join(ROOT, "gpurun_out", f"pmcmem_{tag}_1.txt"), "FETCH_SIZE")
write_kb = counter(os.path.join(ROOT, "gpurun_out", f"pmcmem_{tag}_2.txt"), "WRITE_SIZE")
lib = os.path.join(ROOT, "tacotron2_amd", "libgl_hip.so")
B, T, F, L = 64, 861, 1025, 861 * 256 + 2048
rec = {
    "source": "rocprofv3 --pmc passes of `python3 bench.py --steps 1 --warmup 0 --preroll-ms 0 --cpu-seconds 0 --graph 0` "
              "(tools_pmc_mem.sh), kernel k_gl2048_regring<2>, average over its 60 dispatches",
    "FETCH_SIZE_KB_per_launch": fetch_kb,
    "WRITE_SIZE_KB_per_launch": write_kb,
    "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE counts 128-byte requests as 64 bytes on gfx950 -> x2 on "
                  "reads; WRITE_SIZE exact",
    "hbm_bytes_per_launch": int(round((2.0 * fetch_kb + write_kb) * 1024.0)),
    "algorithmic_bytes_per_launch": B * (T * F * 4 + 2 * L * 4),
    "lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(),
    "git_head": subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip(),
    "files": [f"profiles/{tag}_pmc_mem_pass1.txt", f"profiles/{tag}_pmc_mem_pass2.txt"],
}
with open(os.path.join(ROOT, "profiles", "r2_traffic.json"), "w") as fh:
    json.dump(rec, fh, indent=1)
print(json.dumps(rec, indent=1))
