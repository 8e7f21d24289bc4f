bash tools/final_pack.sh r6/final3
