"""Turn a tools/profile.sh output directory (gpurun_out/<tag>) into the committed evidence under profiles/<name>/:
kernel_stats_<wl>.csv (rocprofv3 --kernel-trace --stats), traffic_<wl>.json (HBM bytes per launch from the FETCH_SIZE /
WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes: both counters are in KB; on gfx950 FETCH_SIZE reports half the
bytes of wide coalesced streaming reads, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane streaming stores),
bench_<wl>.json, and profiles/traffic_latest.json (read by bench.py for the roofline `traffic` field).

    python tools/summarize_profile.py gpurun_out/p7 profiles/r01_v3
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys


def short(name):
    name = name.replace("void mgacbam::", "").replace("mgacbam::", "")
    return re.split(r"[<(]", name)[0]


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    latest = {}
    for wl in ("cfg2", "cfg4", "cfg3", "cfg1"):
        tr = glob.glob(os.path.join(src, f"trace_{wl}", "*", "*_kernel_stats.csv"))
        if not tr:
            continue
        shutil.copy(tr[0], os.path.join(dst, f"kernel_stats_{wl}.csv"))
        stats = {}
        for r in csv.DictReader(open(tr[0])):
            if "mgacbam" in r["Name"]:
                stats[short(r["Name"])] = dict(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3,
                                               min_us=float(r["MinNs"]) / 1e3, max_us=float(r["MaxNs"]) / 1e3, full_name=r["Name"])
        traffic = collections.defaultdict(dict)
        for ctr, key in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
            f = glob.glob(os.path.join(src, f"pmc_{ctr}_{wl}", "*", "*_counter_collection.csv"))
            if not f:
                continue
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f[0])):
                if "mgacbam" in r["Kernel_Name"] and r["Counter_Name"] == key:
                    agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
            for k, v in agg.items():
                traffic[k][key + "_KB_raw_mean"] = sum(v) / len(v)
        out = {}
        for k, t in traffic.items():
            fetch = t.get("FETCH_SIZE_KB_raw_mean", 0.0) * 1024 * 2      # gfx950: FETCH_SIZE = 1/2 of wide coalesced reads
            write = t.get("WRITE_SIZE_KB_raw_mean", 0.0) * 1024
            out[k] = dict(hbm_read_bytes=round(fetch), hbm_write_bytes=round(write), hbm_bytes=round(fetch + write),
                          **{kk: round(vv, 1) for kk, vv in t.items()}, **stats.get(k, {}))
        json.dump(out, open(os.path.join(dst, f"traffic_{wl}.json"), "w"), indent=1, sort_keys=True)
        latest[wl] = {k: v["hbm_bytes"] for k, v in out.items()}
        b = os.path.join(src, f"bench_{wl}.json")
        if os.path.exists(b):
            lines = [l for l in open(b) if l.startswith("{")]
            if lines:
                open(os.path.join(dst, f"bench_{wl}.json"), "w").write(lines[-1])
        print(wl)
        for k, v in sorted(out.items()):
            print(f"  {k:16s} avg {v.get('avg_us', 0):7.2f} us   HBM read {v['hbm_read_bytes'] / 1e6:8.1f} MB  write {v['hbm_write_bytes'] / 1e6:8.1f} MB")
    # traffic_latest.json keeps the newest figures per workload and says where each came from (bench.py quotes it as `traffic_source`)
    lp = os.path.join(os.path.dirname(os.path.abspath(dst)), "traffic_latest.json")
    merged = json.load(open(lp)) if os.path.exists(lp) else {}
    try:
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip()
    except Exception:
        commit = "unknown"
    src_tag = dict(merged.get("_source", {})) if isinstance(merged.get("_source"), dict) else {}
    for wl, v in latest.items():
        merged[wl] = v
        src_tag[wl] = f"{os.path.basename(os.path.normpath(dst))} @ {commit}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (separate runs), FETCH_SIZE x2 (gfx950)"
    merged["_source"] = src_tag
    json.dump(merged, open(lp, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
