#!/bin/bash
# builds build/lab/gemm_lab_<name> for every ablation set given as "name:-Dflags" (development aid, round 5)
set -euo pipefail
cd "$(dirname "$0")/../.."
mkdir -p build/lab
EXTRA='case 40: return launch_one<MM, OT, EP, 64, 128, 32, 32, 3, LD_OV, 128>(p, s); case 41: return launch_one<MM, OT, EP, 64, 128, 32, 32, 4, LD_OV, 128>(p, s); case 42: return launch_one<MM, OT, EP, 64, 128, 32, 32, 5, LD_OV, 128>(p, s); case 43: return launch_one<MM, OT, EP, 128, 128, 64, 32, 3, LD_OV, 128>(p, s); case 44: return launch_one<MM, OT, EP, 256, 128, 64, 64, 3, LD_OV, 128>(p, s); case 46: return launch_one<MM, OT, EP, 256, 160, 32, 160, 3, LD_PIPE, 128>(p, s); case 47: return launch_one<MM, OT, EP, 256, 160, 32, 160, 3, LD_DMA, 128>(p, s); case 50: return launch_one<MM, OT, EP, 256, 160, 32, 160, 3, LD_OG, 128>(p, s); case 51: return launch_one<MM, OT, EP, 256, 128, 64, 64, 3, LD_OG, 128>(p, s); case 52: return launch_one<MM, OT, EP, 128, 128, 64, 32, 3, LD_OG, 128>(p, s); case 53: return launch_one<MM, OT, EP, 128, 128, 64, 32, 4, LD_OG, 128>(p, s); case 45: return launch_one<MM, OT, EP, 128, 128, 64, 32, 4, LD_OV, 128>(p, s);'
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSDNQ_PRELOAD_GEMM -mllvm -amdgpu-kernarg-preload-count=14 \
     "-DLAB_EXTRA=$EXTRA" $flags tools/micro/gemm_lab.hip -o build/lab/gemm_lab_$name &
done
wait
ls -la build/lab
