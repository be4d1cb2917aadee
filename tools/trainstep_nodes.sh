#!/bin/bash
# Kernel list of one replayed cfg-3 train step with start/end times (run through gpurun).  NOTE: under the tracer every
# dependent kernel shows >= 4.5 us; un-profiled A/Bs (tools/ab_trainstep.py style, one box) showed that removing such a
# node saves ~2 us, not 5, and that parallel graph branches cost ~10 us per cross-queue dependency.
#   bash tools/trainstep_nodes.sh <tag>   -> gpurun_out/<tag>/trainstep_nodes.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1/nodes; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 tools/bench_trainstep.py --graph --steps 20 --warmup 3 > $O/run.log 2>&1
python3 - <<PY > $R/gpurun_out/$1/trainstep_nodes.txt
import csv, glob
f=glob.glob('$O/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last complete step: from the second-to-last fused-Adam kernel to the last one
adam=[i for i,r in enumerate(rows) if 'FusedOptimizer' in r['Kernel_Name']]
a,b=adam[-2],adam[-1]
t0=int(rows[a]['Start_Timestamp'])
print('one replayed step, us from the start of the previous Adam kernel: start, end, duration, queue, kernel')
for r in rows[a:b+1]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:60]
    print('%8.1f %8.1f %7.1f  q%s  %s' % ((s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3, r.get('Queue_Id','?'), n))
print('step period %.1f us, %d kernels' % ((int(rows[b]['Start_Timestamp'])-t0)/1e3, b-a))
PY
cat $R/gpurun_out/$1/trainstep_nodes.txt
rm -rf $O
