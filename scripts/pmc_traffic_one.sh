# usage: bash scripts/pmc_traffic_one.sh <one_conv args...>   -> FETCH_SIZE / WRITE_SIZE (KiB) per launch
R=$PWD; cd /tmp; export TMPDIR=/tmp
for P in FETCH_SIZE WRITE_SIZE; do
  D=$R/gpurun_out/pmc_t_$P
  rm -rf $D
  timeout -k 10 120 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -- python3 $R/scripts/one_conv.py "$@" > $D.log 2>&1 || echo "pass failed: $P"
done
cd $R; for d in gpurun_out/pmc_t_FETCH_SIZE gpurun_out/pmc_t_WRITE_SIZE; do python3 scripts/pmc_summary.py $d conv; done | awk '{print $(NF-7), $(NF-6), $(NF-5), $(NF-4), $(NF-3), $(NF-2), $(NF-1), $NF}'
