"""Condenses rocprofv3 output (gpurun_out/<tag>_trace, _pmc_fetch, _pmc_write) into profiles/.

usage: python scripts/summarize_profile.py r1
Writes profiles/<tag>_kernel_stats.csv (per-kernel durations of the `--kernel-trace --stats` run),
profiles/<tag>_traffic.json and profiles/traffic.json (HBM bytes per launch of the GEMM kernels,
from the separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes) and, where a `<tag>_pmc_mfma` pass
exists, profiles/<tag>_mfma.json and profiles/mfma.json (matrix-core utilisation per GEMM kernel from
SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CU_CYCLES).

Counter conventions (MI355X): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a
wide coalesced streaming read (16 B per lane, which is what these kernels issue) at exactly half
its bytes, so the read side is doubled; WRITE_SIZE is exact.
"""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
out_dir = os.path.join(ROOT, "profiles")
os.makedirs(out_dir, exist_ok=True)


def demangle(name):
    if not name.startswith("_Z"):
        return name
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool, name], capture_output=True, text=True, check=True).stdout.strip()
        except (OSError, subprocess.CalledProcessError):
            continue
        if out and out != name:
            return out
    return name


def short(name):
    name = demangle(name)
    m = re.match(r"_ZN\d+_GLOBAL__N_1\d+(\w+?_kernel)I((?:Li\d+E)+)E", name)  # no demangler knew the _Float16 mangling
    if m:
        return m.group(1) + "<" + ", ".join(re.findall(r"Li(\d+)E", m.group(2))) + ">"
    name = name.replace("(anonymous namespace)::", "")
    if name.startswith("void "):
        name = name[5:]
    m = re.search(r"(gemm_\w+_kernel<[^>]*>|dequant_\w+_kernel|repack_\w+_kernel|stream_read_kernel)", name)
    if m:
        return m.group(1)
    name = re.sub(r"\(.*", "", name)
    return name[:80]


def one(pattern):
    files = glob.glob(os.path.join(ROOT, "gpurun_out", pattern), recursive=True)
    if not files:
        raise SystemExit(f"missing {pattern}")
    return files[0]


rows = list(csv.DictReader(open(one(f"{tag}_trace/**/*kernel_stats.csv"))))
with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "avg_ns", "min_ns", "max_ns", "stddev_ns", "pct_of_gpu_time"])
    for r in rows[:20]:
        w.writerow([short(r["Name"]), r["Calls"], f"{float(r['AverageNs']):.0f}", r["MinNs"], r["MaxNs"],
                    f"{float(r['StdDev']):.0f}", r["Percentage"]])


def per_kernel(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        if not k.startswith("gemm_"):
            continue
        acc.setdefault(k, []).append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


fetch = per_kernel(one(f"{tag}_pmc_fetch/**/*counter_collection.csv"), "FETCH_SIZE")
write = per_kernel(one(f"{tag}_pmc_write/**/*counter_collection.csv"), "WRITE_SIZE")
traffic = {"_note": "HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024; separate --pmc passes; "
                    "FETCH_SIZE doubled per the gfx950 half-count of 16-byte-per-lane streaming reads",
           "kernels": {}}
for k in sorted(fetch):
    f_kib, n = fetch[k]
    w_kib = write.get(k, (0.0, 0))[0]
    traffic["kernels"][k] = {"launches": n, "FETCH_SIZE_KiB": f_kib, "WRITE_SIZE_KiB": w_kib,
                             "hbm_bytes_per_launch": (2 * f_kib + w_kib) * 1024}
# the two launches bench.py reports rooflines for, under the names the library gives them
sys.path.insert(0, ROOT)
from vllm_quick_amd._lib import launch_shape  # noqa: E402  (host arithmetic only)

for key, m in (("m128", 128), ("m1", 1)):
    want = launch_shape(m, 4096, 11008)
    for k, v in traffic["kernels"].items():
        if k.replace(" ", "").startswith(want[:-1] + ","):
            traffic[f"{key}_hbm_bytes_per_launch"] = v["hbm_bytes_per_launch"]
            traffic[f"{key}_kernel"] = want
traffic["commit"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True,
                                   text=True).stdout.strip()
traffic["dirty"] = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "vllm_quick_amd/csrc"],
                                       capture_output=True, text=True).stdout.strip())
traffic["source"] = f"profiles/{tag}_traffic.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE on bench.py)"
for name in (f"{tag}_traffic.json", "traffic.json"):
    with open(os.path.join(out_dir, name), "w") as f:
        json.dump(traffic, f, indent=1)
# ---- matrix-core utilisation from counters (its own --pmc pass) ----
# SQ_VALU_MFMA_BUSY_CYCLES: cycles a SIMD's matrix pipe is busy, summed over all SIMDs (32 per v_mfma_f32_32x32x16_f16,
# 16 per 16x16x32).  SQ_BUSY_CU_CYCLES: cycles a CU holds at least one wave, summed over all CUs.  A CU has 4 SIMDs:
#   mfma_busy = MFMA_BUSY / (4 * BUSY_CU)     share of the matrix pipes' time in use while their CU is busy
# The counters tick with the clock the chip actually holds (it drops to ~1.6-2.0 GHz under matrix load), so this
# figure is higher than flops / time / (the 2.5 PFLOP/s the data sheet quotes at 2.4 GHz) by that clock ratio and by
# the share of CUs a launch leaves idle.
mfma_files = glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_mfma/**/*counter_collection.csv"), recursive=True)
if mfma_files:
    busy = per_kernel(mfma_files[0], "SQ_VALU_MFMA_BUSY_CYCLES")
    cu = per_kernel(mfma_files[0], "SQ_BUSY_CU_CYCLES")
    mfma = {"_note": "per launch; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES): share of the "
                     "matrix pipes' time in use while their CU holds a wave (own --pmc pass)",
            "kernels": {}}
    for k in sorted(busy):
        if k not in cu or cu[k][0] <= 0:
            continue
        mfma["kernels"][k] = {"launches": busy[k][1], "SQ_VALU_MFMA_BUSY_CYCLES": busy[k][0],
                              "SQ_BUSY_CU_CYCLES": cu[k][0], "mfma_busy": busy[k][0] / (4.0 * cu[k][0])}
    for m in (32, 128, 512, 1024, 4096):
        want = launch_shape(m, 4096, 11008)
        for k, v in mfma["kernels"].items():
            if k.replace(" ", "").startswith(want[:-1] + ","):
                mfma[f"m{m}_kernel"] = want
                mfma[f"m{m}_mfma_busy"] = v["mfma_busy"]
                mfma[f"m{m}_mfma_busy_cycles"] = v["SQ_VALU_MFMA_BUSY_CYCLES"]
    mfma["commit"], mfma["dirty"] = traffic["commit"], traffic["dirty"]
    mfma["source"] = (f"profiles/{tag}_mfma.json (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES on "
                      "bench.py --sweep-ms 32,128,512,1024,4096)")
    for name in (f"{tag}_mfma.json", "mfma.json"):
        with open(os.path.join(out_dir, name), "w") as f:
            json.dump(mfma, f, indent=1)
    print(json.dumps(mfma, indent=1))
for src, dst in ((f"{tag}_bench.json", f"{tag}_bench.json"), (f"{tag}_bench_under_rocprof.json", f"{tag}_bench_under_rocprof.json")):
    if os.path.exists(os.path.join(ROOT, "gpurun_out", src)):
        shutil.copyfile(os.path.join(ROOT, "gpurun_out", src), os.path.join(out_dir, dst))
print(json.dumps(traffic, indent=1))
