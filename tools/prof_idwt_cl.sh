#!/bin/bash
# HBM bytes, SQ / L1 / L2 / address-translation counters of the channel-last level kernels (and the channel-first synthesis) (run through gpurun from the repository root):
#   bash tools/prof_idwt_cl.sh <tag> [d]   -> gpurun_out/<tag>/idwt_cl_pmc.txt
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; D=${2:-65}
mkdir -p $O; cd $R
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_RDREQ_32B_sum" \
           "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/clpmc_$i -- python3 tools/microbench/idwt_cl_probe.py $D > $O/clpmc_$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections, re
O='$O'
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(O+'/clpmc_*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        m=re.search(r'(idwt_\w+|analysis_\w+)(<[^>]*>)?', r['Kernel_Name'])
        if m: acc[m.group(0)][r['Counter_Name']].append(float(r['Counter_Value']))
with open(O+'/idwt_cl_pmc.txt','w') as out:
    for k,v in acc.items():
        out.write(k+'\n'); print(k)
        for c,x in sorted(v.items()):
            line='   %-34s %.4g' % (c, sum(x)/len(x)); out.write(line+'\n'); print(line)
PY
