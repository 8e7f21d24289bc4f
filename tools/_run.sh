mkdir -p gpurun_out/r6
timeout 1200 python -m pytest tests/test_capture.py tests/test_hf_plugin.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r6/pytest_capture.txt
cat gpurun_out/r6/pytest_capture.txt
O=gpurun_out/r6/launch_modes.txt; : > $O
for i in 1 2; do
for mode in graph capture eager; do
ms=$(python bench.py --steps 20 --warmup 3 --no-cpu-baseline --launch $mode 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "|", d["config"]["launch"][:60])')
echo "launch=$mode: $ms" | tee -a $O
done
done
