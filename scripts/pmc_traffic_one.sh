R=$PWD; cd /tmp; export TMPDIR=/tmp
for P in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  D=$R/gpurun_out/pmc_t_$(echo $P | cut -d' ' -f1)
  rm -rf $D
  timeout -k 10 120 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $D -- python3 $R/scripts/one_conv.py 52 128 256 3 1 same 32 5 fwdp > $D.log 2>&1 || echo "pass failed: $P"
done
cd $R; for d in gpurun_out/pmc_t_*/; do python3 scripts/pmc_summary.py $d conv; done | awk '{print $(NF-7), $(NF-6), $(NF-5), $(NF-4), $(NF-3), $(NF-2), $(NF-1), $NF}'
