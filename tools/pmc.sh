#!/bin/bash
# Developer tool (GPU box): PMC passes over tools/run_jac.py; prints per-kernel sums.
# usage: tools/pmc.sh "<emit spec>" <outdir> [jac|con|fused]
SPEC="$1"; OUT="$GRAFT_REPO_ROOT/gpurun_out/$2"; WHAT="${3:-jac}"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INSTS_VALU"
 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD"
 "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR"
 "FETCH_SIZE GRBM_GUI_ACTIVE"
 "WRITE_SIZE"
)
i=0
for P in "${PASSES[@]}"; do
  rocprofv3 --pmc $P -d "$OUT/p$i" -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/tools/run_jac.py "$SPEC" $WHAT > "$OUT/p$i.log" 2>&1
  i=$((i+1))
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0]
        acc[k][row['Counter_Name']] += float(row['Counter_Value'])
        cnt[(k, row['Counter_Name'])] += 1
for k in acc:
    if not k.startswith('opty_'): continue
    print(k)
    for c, v in sorted(acc[k].items()):
        n = cnt[(k, c)]
        print('   %-28s %16.0f per-dispatch (n=%d)' % (c, v/n, n))
PY
