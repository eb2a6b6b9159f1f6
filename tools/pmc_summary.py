#!/usr/bin/env python3
"""Summarise the rocprofv3 PMC passes of tools/pmc.sh into profiles/<tag>_traffic.json.

HBM bytes per launch of every step kernel = FETCH_SIZE x calibrated factor + WRITE_SIZE (KiB units, separate
passes), calibrated with hs_debug_calibrate's dword-per-lane copy as MI355X_MICROARCH.md prescribes.
usage: pmc_summary.py <gpurun_out dir> <out json> [worlds]"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def read(dirpat):
    per = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(dirpat + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            per[k][0] += 1
            per[k][1] += float(r["Counter_Value"])
    return per


def short(name):
    m = re.search(r"hs::(k_[a-z_]+)(<[^>]*>)?", name)
    if not m:
        return None
    return m.group(1)


def main():
    root, out = sys.argv[1], sys.argv[2]
    worlds = int(sys.argv[3]) if len(sys.argv) > 3 else 16000
    fetch, write = read(root + "/pmc_fetch"), read(root + "/pmc_write")
    cf, cw = read(root + "/pmc_calib_fetch"), read(root + "/pmc_calib_write")
    calib_bytes = float(1 << 29)
    kf = [v for k, v in cf.items() if "calib" in k][0]
    kw = [v for k, v in cw.items() if "calib" in k][0]
    fetch_factor = calib_bytes / (kf[1] / kf[0] * 1024.0)
    write_factor = calib_bytes / (kw[1] / kw[0] * 1024.0)
    per = {}
    for name, (n, tot) in fetch.items():
        s = short(name)
        if s is None:
            continue
        wn, wtot = write.get(name, [n, 0.0])
        per[s] = {"launches": n, "fetch_KiB_per_launch_raw": tot / n, "write_KiB_per_launch": wtot / max(wn, 1),
                  "hbm_bytes_per_launch": tot / n * 1024.0 * fetch_factor + wtot / max(wn, 1) * 1024.0 * write_factor}
    phys = per.get("k_physics", {}).get("hbm_bytes_per_launch")       # one persistent kernel per step
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_fingerprint
    res = {"csrc_sha": csrc_fingerprint(),          # the kernel sources this run measured (bench.py drops a stale file)
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc.sh), bench.py --steps 40, %d worlds; summarised by tools/pmc_summary.py" % worlds,
           "calibration": {"pattern": "one coalesced dword per lane, 512 MiB read + 512 MiB written (hs_debug_calibrate)",
                           "fetch_factor": fetch_factor, "write_factor": write_factor},
           "per_kernel": per,
           "physics": {"hbm_bytes_per_step": phys, "algorithmic_bytes_per_step": 1880.0 * worlds},
           "observe": {"hbm_bytes_per_step": per.get("k_observe", {}).get("hbm_bytes_per_launch"), "algorithmic_bytes_per_step": 6300.0 * worlds},
           "reset": {"hbm_bytes_per_step": per.get("k_reset", {}).get("hbm_bytes_per_launch")}}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("calibration", "physics", "observe", "reset")}))


if __name__ == "__main__":
    main()
