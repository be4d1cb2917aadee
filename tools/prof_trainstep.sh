set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2/ts_$1
mkdir -p $O
cd $R
python tools/bench_trainstep.py --graph --steps 100 --warmup 10 > $O/plain.log 2>&1 || true
tail -2 $O/plain.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 tools/bench_trainstep.py --graph --steps 40 --warmup 5 > $O/prof.log 2>&1 || true
f=$(ls $O/prof/*/*kernel_stats.csv | head -1)
cp $f $O/trainstep_kernel_stats.csv
head -25 $O/trainstep_kernel_stats.csv | cut -c1-200
