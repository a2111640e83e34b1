# usage (on the GPU box, from the repo root): bash scripts/round_profile.sh <tag>   e.g. r01_h
# Writes gpurun_out/<tag>_*: the bench line, rocprofv3 kernel stats (two streams / one stream) and the two PMC passes.
TAG=$1; R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py --steps 10 --warmup 3 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.log || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_ks -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-extra-blocks > $O/${TAG}_ks.log 2>&1 || exit 2
cp $O/${TAG}_ks/*/*kernel_stats.csv $O/${TAG}_bench_kernel_stats.csv; rm -rf $O/${TAG}_ks
export YOLO_BWD_OVERLAP=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_ks1 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-extra-blocks > $O/${TAG}_ks1.log 2>&1 || exit 3
cp $O/${TAG}_ks1/*/*kernel_stats.csv $O/${TAG}_serial_kernel_stats.csv; rm -rf $O/${TAG}_ks1
unset YOLO_BWD_OVERLAP
for P in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/${TAG}_pmc_$P -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-extra-blocks > $O/${TAG}_pmc_$P.log 2>&1 || exit 4
done
cd $R
python3 scripts/pmc_traffic.py $O/${TAG}_pmc_FETCH_SIZE $O/${TAG}_pmc_WRITE_SIZE $O/${TAG}_hbm_traffic.json
rm -rf $O/${TAG}_pmc_FETCH_SIZE $O/${TAG}_pmc_WRITE_SIZE
cat $O/${TAG}_bench.json
