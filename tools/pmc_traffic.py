#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KB units per dispatch).
usage: pmc_traffic.py <dir_fetch> <dir_write> [name-substring]  -> prints JSON {kernel: {launches, fetch_bytes, write_bytes}}
gfx950 note (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads -> doubled here."""
import collections, csv, glob, json, sys
def load(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc
filt = sys.argv[3] if len(sys.argv) > 3 else "gemm_kernel"
fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
tot_f = tot_w = n = 0
for k in fe:
    if filt not in k:
        continue
    f, w = fe[k], wr.get(k, [0.0])
    out[k[:110]] = {"launches": len(f), "fetch_bytes_per_launch_corrected": 2 * 1024 * sum(f) / len(f), "write_bytes_per_launch": 1024 * sum(w) / max(1, len(w))}
    tot_f += 2 * 1024 * sum(f); tot_w += 1024 * sum(w); n += len(f)
out["_all_" + filt] = {"launches": n, "hbm_bytes_per_launch": (tot_f + tot_w) / max(1, n), "fetch_bytes_per_launch_corrected": tot_f / max(1, n), "write_bytes_per_launch": tot_w / max(1, n)}
import datetime, os, subprocess
def _git(*a):
    try:
        return subprocess.run(["git", *a], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip()
    except OSError:
        return ""
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sdnq_amd", "libsdnq_hip.so.srchash")
out["_provenance"] = {"collected_utc": datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%dT%H:%MZ"),
                      "library_srchash": open(lib).read().strip()[:16] if os.path.exists(lib) else None,
                      "method": "two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/pmc_shapes.py; FETCH_SIZE doubled (gfx950 correction)"}
print(json.dumps(out, indent=1))
