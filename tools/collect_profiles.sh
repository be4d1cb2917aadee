#!/bin/bash
# Collects the round's profile evidence on the GPU box (run through gpurun from the repository root):
#   bash tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>/...   (copy what is to be judged into profiles/<round>/)
# rocprofv3 runs: --kernel-trace --stats alone; every --pmc set in a run of its own (FETCH_SIZE and WRITE_SIZE cannot
# share a pass); the profiled program is always `python3 <script>` directly after `--`.
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
echo "== default bench run"; python3 bench.py > $O/bench_default_run.log 2>&1; tail -c 400 $O/bench_default_run.log; echo
# --stats averages over EVERY launch, warm-up included (the first launches of a process run 5-25 % long); the summary beside
# it is made from the per-dispatch trace with the first WARM launches of each kernel dropped: median, mean, min, max.  That
# mean is what bench.py's roofline.kernel_ms (HIP events over the timed steps only) has to agree with.
summarize() {  # <kernel_trace.csv> <out.csv> <warm-up launches to drop>
python3 - "$1" "$2" "$3" <<'PY'
import csv, sys, statistics as st, collections
src, dst, warm = sys.argv[1], sys.argv[2], int(sys.argv[3])
d = collections.defaultdict(list)
for r in csv.DictReader(open(src)):
    d[r['Kernel_Name']].append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
rows = []
for k, v in d.items():
    v.sort()
    t = [x[1] for x in v][warm if len(v) > warm else 0:]
    rows.append((sum(t), k, len(v), len(t), st.median(t), sum(t) / len(t), min(t), max(t)))
with open(dst, 'w') as fh:
    w = csv.writer(fh)
    w.writerow(['Kernel_Name', 'Calls', 'Calls_after_warmup', 'Median_ns', 'Mean_after_warmup_ns', 'Min_ns', 'Max_ns'])
    for r in sorted(rows, reverse=True):
        w.writerow([r[1][:160]] + list(r[2:]))
PY
}
for prec in f16x2 fp32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$prec -- python3 bench.py --precision $prec --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $O/kt_$prec.log 2>&1
  cp $(ls $O/kt_$prec/*/*kernel_stats.csv | head -1) $O/bench_headline_${prec}_kernel_stats.csv
  summarize $(ls $O/kt_$prec/*/*kernel_trace.csv | head -1) $O/bench_headline_${prec}_kernel_summary.csv 2
  grep -o '"kernel_ms": [0-9.]*' $O/kt_$prec.log | head -1 > $O/bench_headline_${prec}_kernel_ms_of_that_run.txt
done
for wl in cfg2 cfg5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$wl -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/kt_$wl.log 2>&1
  cp $(ls $O/kt_$wl/*/*kernel_stats.csv | head -1) $O/bench_${wl}_f16x2_kernel_stats.csv
  summarize $(ls $O/kt_$wl/*/*kernel_trace.csv | head -1) $O/bench_${wl}_f16x2_kernel_summary.csv 1
done
echo "== PMC passes (headline)"
i=0
for prec in f16x2 fp32; do
  for set in "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH SQ_INSTS_VALU_MFMA_MOPS_F16"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$i -- python3 bench.py --precision $prec --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $O/pmc_$i.log 2>&1
    echo "$prec|$set" > $O/pmc_$i/what.txt
  done
done
python3 - <<PY
import csv, glob, json, collections, os
O='$O'
res={'f16x2':{}, 'fp32':{}}
for d in sorted(glob.glob(O+'/pmc_*/')):
    what=open(d+'what.txt').read().strip(); prec=what.split('|')[0]
    for f in glob.glob(d+'*/*counter_collection.csv'):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'lfgc_fwd16_kernel' in r['Kernel_Name'] or ('lfgc_fwd_kernel' in r['Kernel_Name'] and prec=='fp32'):
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items():
            res[prec][k]=sum(v)/len(v)
out={'commands': 'rocprofv3 --pmc <one set> --kernel-trace --output-format csv -- python3 bench.py [--precision fp32] --steps 2 --warmup 1 --no-cpu-baseline --no-extra; one run per set (FETCH_SIZE and WRITE_SIZE in runs of their own); values = mean over the launches of the dominant kernel',
     'workload': 'headline: 256^3 lattice, 64^3x32ch grid, MLP 4x128; one launch = 16,777,216 samples = 524,288 tiles of 32',
     'hbm': {}, 'sq_counters': res,
     'note': 'gfx950: FETCH_SIZE counts half of 16-B/lane reads -> HBM bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (MI355X_MICROARCH.md, HBM section); SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES in cycles, GRBM_GUI_ACTIVE summed over the 8 XCDs'}
for prec in res:
    if 'FETCH_SIZE' in res[prec] and 'WRITE_SIZE' in res[prec]:
        out['hbm'][prec]={'FETCH_SIZE_KB':res[prec]['FETCH_SIZE'],'WRITE_SIZE_KB':res[prec]['WRITE_SIZE'],
                          'hbm_bytes_per_launch': int(2*res[prec]['FETCH_SIZE']*1024+res[prec]['WRITE_SIZE']*1024),
                          'algorithmic_bytes_per_launch': 16777216*1040}
json.dump(out, open(O+'/bench_headline_pmc.json','w'), indent=1)
print(json.dumps(out['hbm']))
for prec in res:
    print(prec, {k: round(v) for k,v in res[prec].items()})
PY
echo "== train step (graph replay)"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_ts -- python3 tools/bench_trainstep.py --graph --steps 40 --warmup 5 > $O/kt_ts.log 2>&1
cp $(ls $O/kt_ts/*/*kernel_stats.csv | head -1) $O/trainstep_cfg3_graph_kernel_stats.csv
summarize $(ls $O/kt_ts/*/*kernel_trace.csv | head -1) $O/trainstep_cfg3_graph_kernel_summary.csv 8
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_ts_sm -- python3 tools/bench_trainstep.py --graph --steps 40 --warmup 5 --drop-type smallify > $O/kt_ts_sm.log 2>&1
cp $(ls $O/kt_ts_sm/*/*kernel_stats.csv | head -1) $O/trainstep_cfg3_smallify_graph_kernel_stats.csv
python3 tools/bench_trainstep.py --graph --steps 100 --warmup 10 > $O/trainstep_plain.log 2>&1; tail -1 $O/trainstep_plain.log | cut -c1-200
echo "== microbenchmarks"
python3 tools/microbench/idwt_sizes.py > $O/idwt_sizes.log 2>&1; tail -4 $O/idwt_sizes.log
python3 tests/error_stats.py > $O/fwd_error_stats.log 2>&1
LFGC_AB_DIR=$O/st python3 tools/phase_stamps.py build > $O/stamps_build.log 2>&1 && LFGC_AB_DIR=$O/st python3 tools/phase_stamps.py run > $O/phase_stamps.log 2>&1 && LFGC_AB_DIR=$O/st python3 tools/phase_stamps.py run bwd > $O/phase_stamps_bwd.log 2>&1
rm -rf $O/st
echo "== sharded driver: host overhead on an 8-rank share (RCCL, world size 1)"
python3 tools/sharded_host_overhead.py 2>&1 | grep -v "^\[W\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" > $O/sharded_host_overhead.log; cat $O/sharded_host_overhead.log
# (the channel-last wavelet level kernels are unchanged since round 2: their counters / ablations stay in profiles/r2)
rm -rf $O/kt_*/ $O/pmc_*/
ls $O
