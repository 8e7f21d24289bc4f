for v in "" build/exp/lib_noload.so build/exp/lib_nolds.so; do
  echo "LIB=$v"; SDNQ_HIP_LIB=$PWD/$v; [ -z "$v" ] && unset SDNQ_HIP_LIB || export SDNQ_HIP_LIB
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pc; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o c -- python $GRAFT_REPO_ROOT/bench.py --workload sdxl_conv_int8 --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; grep -E "conv_quant|gemm_kernel" /tmp/pc/c_kernel_stats.csv | cut -c1-60,150-260; cd $GRAFT_REPO_ROOT
done
