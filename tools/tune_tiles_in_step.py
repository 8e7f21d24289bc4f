#!/usr/bin/env python3
"""Development aid: judge GEMM tiles by the STEP, not by a kernel replayed alone.  For every listed problem of a bench.py workload,
force each candidate tile for that problem only (SDNQ_HIP_TILE_MAP) and run the bench; prints ms per step against the heuristics' choice.
usage: tools/tune_tiles_in_step.py [workload] [steps]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
workload = sys.argv[1] if len(sys.argv) > 1 else "sdxl_int8"
steps = sys.argv[2] if len(sys.argv) > 2 else "20"
SHAPES = {"sdxl_int8": [(4096, 640, 640), (4096, 1920, 640), (4096, 5120, 640), (4096, 640, 2560), (1024, 1280, 1280), (1024, 3840, 1280),
                        (1024, 10240, 1280), (1024, 1280, 5120)]}
SHAPES["sdxl_int8_dequant"] = [(m, n, 2 * k) for (m, n, k) in SHAPES["sdxl_int8"]]  # the fused dequantize GEMM counts K in bytes of the 16-bit rows
SHAPES["flux_int4_had"] = [(4608, 3072, 3072), (4608, 3072, 12288), (4608, 3072, 15360), (512, 3072, 3072), (512, 3072, 12288)]
SHAPES = SHAPES.get(workload, [])
CANDS = {"sdxl_int8": [1, 3, 7, 9, 10, 13, 17, 19], "flux_int4_had": [1, 3, 19, 20]}.get(workload, [1, 2, 3, 4])  # launch_tiles / launch_tiles_w8 ids


if os.environ.get("TUNE_SHAPES"):  # "MxNxK,MxNxK": only these problems
    SHAPES = [tuple(int(v) for v in t.split("x")) for t in os.environ["TUNE_SHAPES"].split(",")]
if os.environ.get("TUNE_CANDS"):   # "1,21,22": only these tile ids
    CANDS = [int(v) for v in os.environ["TUNE_CANDS"].split(",")]


def run(env_map):
    env = dict(os.environ)
    if env_map:
        env["SDNQ_HIP_TILE_MAP"] = env_map
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", steps, "--warmup", "3", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env)
    try:
        return json.loads(r.stdout.strip().splitlines()[-1])["ms_per_step"]
    except Exception:
        return None


base = [run(None) for _ in range(3)]
print("heuristics:", base, flush=True)
b = min(x for x in base if x)
for (m, n, k) in SHAPES:
    line = f"{m}x{n}x{k}:"
    for t in CANDS:
        if t == 13 and n % 160:
            continue
        ms = run(f"{m}x{n}x{k}={t}")
        line += f"  {t}: " + (f"{ms:.3f} ({ms - b:+.3f})" if ms else "fail")
    print(line, flush=True)
if os.environ.get("TUNE_ALL"):  # every listed problem on one candidate at once
    for t in CANDS:
        ms = run(",".join(f"{m}x{n}x{k}={t}" for (m, n, k) in SHAPES))
        print(f"all listed problems on tile {t}: " + (f"{ms:.3f} ({ms - b:+.3f})" if ms else "fail"), flush=True)
print("heuristics again:", run(None), flush=True)
