for seed in 7 8 9; do timeout 900 python tools/fuzz_modes.py $seed 250 2>&1 | grep -i "cover\|skip\|not built\|unsupported" | cut -c1-330 | sort | uniq -c | sort -rn | head -12; done
