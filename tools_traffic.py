#!/usr/bin/env python3
"""profiles/r3_traffic.json from the memory-side PMC passes of tools_pmc_mem.sh <tag>:

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
# K1 on the bench path (k_mel_to_mag<false, true, true>, the PACKED instance gl_mel_run launches): MFMA counters of
# tools_pmc_k1.sh, when that pass has been run for this tag
k1_path = os.path.join(ROOT, "gpurun_out", f"pmck1_{tag}.txt")
if os.path.exists(k1_path):
    k1 = {}
    for line in open(k1_path):
        m = re.match(r"(\S.*?) (SQ_\w+|GRBM_\w+) per-dispatch ([0-9.e+]+) dispatches (\d+)", line.strip())
        if m:
            k1[m.group(2)] = float(m.group(3))
            k1["kernel"] = m.group(1)
    useful = 2.0 * F * 80 * T * B                      # griffin_lim.py:18: (F x 80) . (80 x T) per utterance
    rec["k1"] = {"counters_per_launch": k1, "useful_flops_per_launch": useful,
                 "mfma_flops_issued_per_launch": k1.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512.0,
                 "file": f"profiles/{tag}_pmc_mel_run.txt",
                 "note": "SQ_INSTS_VALU_MFMA_MOPS_F32 counts units of 512 flops; SQ_VALU_MFMA_BUSY_CYCLES / "
                         "(4 x SQ_BUSY_CU_CYCLES): SIMD-cycles with the matrix pipe busy over the SIMD-cycles of the busy CUs (one MFMA 32x32x2 f32 = 64 cycles)"}
    rec["files"].append(f"profiles/{tag}_pmc_mel_run.txt")
    ks = os.path.join(ROOT, "gpurun_out", f"prof_{tag}", "kernel_stats.csv")
    if os.path.exists(ks):
        import csv
        for row in csv.DictReader(open(ks)):
            if "k_mel_to_mag<false, true, true>" in row["Name"]:
                us = float(row["AverageNs"]) / 1e3
                rec["k1"].update({"avg_us": us, "dispatches": int(row["Calls"]), "achieved_TFLOPs": useful / us / 1e6,
                                  "peak_TFLOPs": 157.3, "frac": useful / us / 1e6 / 157.3,
                                  "kernel_trace": f"profiles/{tag}_kernel_stats.csv"})
            if "k_gl2048_regring<2>" in row["Name"]:
                rec["dominant_kernel_avg_us_kernel_trace"] = float(row["AverageNs"]) / 1e3
with open(os.path.join(ROOT, "profiles", "r3_traffic.json"), "w") as fh:
    json.dump(rec, fh, indent=1)
print(json.dumps(rec, indent=1))
