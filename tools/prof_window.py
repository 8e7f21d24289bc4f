#!/usr/bin/env python3
"""Cut the TIMED steps out of a rocprofv3 kernel trace of `bench.py --profile-markers` and summarise them (round 6, verdict item 5a).

bench.py brackets its timed loop with two marker kernels (torch.cuda._sleep -> `spin_kernel`); every dispatch between the END of the
first and the START of the last marker belongs to the K graph-replayed steps -- nothing of the build, the eager warm-up, the roofline
replays (time_gemm_kernel) or torch's RNG kernels.  Calls per step are therefore integers, and "us per launch inside the step" can be
recomputed from the CSV this writes.
usage: prof_window.py <dir with *_kernel_trace.csv> <steps> [out_prefix]"""
import collections, csv, glob, re, sys

d, steps = sys.argv[1], int(sys.argv[2])
out = sys.argv[3] if len(sys.argv) > 3 else d.rstrip("/") + "_window"
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
if not files:
    sys.exit(f"no *kernel_trace.csv under {d}")
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "spin_kernel" in r["Kernel_Name"]]
if len(marks) < 2:
    sys.exit(f"found {len(marks)} marker kernels (need 2): was bench.py run with --profile-markers?")
lo, hi = int(rows[marks[-2]]["End_Timestamp"]), int(rows[marks[-1]]["Start_Timestamp"])
win = [r for r in rows[marks[-2] + 1:marks[-1]] if int(r["Start_Timestamp"]) >= lo and int(r["End_Timestamp"]) <= hi]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|void ", "", n)
    return re.sub(r"\s+", " ", n)[:140]


agg = collections.OrderedDict()
for r in win:
    wgs = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1) // max(1, int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1))
    key = (short(r["Kernel_Name"]), wgs)
    a = agg.setdefault(key, [0, 0])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
span = hi - lo
busy = sum(a[1] for a in agg.values())
with open(out + "_kernels.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "workgroups", "calls_in_window", "calls_per_step", "total_us", "avg_us", "pct_of_kernel_time"])
    for (name, wgs), (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([name, wgs, n, f"{n / steps:g}", f"{ns / 1e3:.1f}", f"{ns / 1e3 / n:.2f}", f"{100.0 * ns / busy:.2f}"])
by_kernel = collections.OrderedDict()
for (name, wgs), (n, ns) in agg.items():
    b = by_kernel.setdefault(name.split("<")[0].split("(")[0], [0, 0])
    b[0] += n
    b[1] += ns
with open(out + "_summary.txt", "w") as f:
    f.write(f"timed window: {steps} steps, {len(win)} dispatches = {len(win) / steps:g} per step; wall {span / 1e3 / steps:.1f} us per step, "
            f"sum of kernel durations {busy / 1e3 / steps:.1f} us per step\n")
    for name, (n, ns) in sorted(by_kernel.items(), key=lambda kv: -kv[1][1]):
        f.write(f"{name[:60]:60s} calls/step {n / steps:8g}  us/step {ns / 1e3 / steps:9.1f}  avg_us {ns / 1e3 / n:8.2f}\n")
print(open(out + "_summary.txt").read())
