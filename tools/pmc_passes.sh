#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/pmc_passes.sh <outdir> <python script + args...>
# One rocprofv3 --pmc pass per counter group (counters are never combined with trace domains).
out=$1; shift
export TMPDIR=/tmp
groups=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM"
 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL"
 "SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS"
 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
 "SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES"
)
i=0
for g in "${groups[@]}"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --pmc $g --output-format csv -d "$out/pass$i" -o p -- python3 "$@" > "$out/pass$i.log" 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/pass*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if k.startswith('occ::') or 'occ::' in k:
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/summary.txt', 'w') as fh:
    for k, d in acc.items():
        fh.write(k + '\n')
        for c, v in sorted(d.items()):
            fh.write(f'   {c:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}\n')
print(open(out + '/summary.txt').read())
PY
