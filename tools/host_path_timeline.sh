#!/bin/bash
# GPU tool: rocprofv3 timeline (kernel dispatches + memory copies) of a few
# jacobian(free) calls of config 3 through the host path -> gpurun_out/hp_tl
root=$PWD
out=$root/gpurun_out/hp_tl
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -- python $root/tools/host_path_trace.py > $out/run.log 2>&1
cd $root
ls $out/*/ | head
python - <<'PY'
import csv, glob
ks = glob.glob('gpurun_out/hp_tl/*/*kernel_trace.csv')[0]
ms = glob.glob('gpurun_out/hp_tl/*/*memory_copy_trace.csv')[0]
ev = []
for r in csv.DictReader(open(ks)):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + r['Kernel_Name'][:28]))
for r in csv.DictReader(open(ms)):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'M %s %s' % (r.get('Direction', r.get('Name', '?'))[:24], r.get('Size', ''))))
ev.sort()
# the last complete call: find the last opty_pack_kernel group of 8
packs = [i for i, e in enumerate(ev) if 'opty_pack' in e[2]]
last = packs[-8]
# start from the H2D copy before
i0 = last
while i0 > 0 and ev[i0][0] - ev[i0 - 1][1] < 200000 and last - i0 < 12:
    i0 -= 1
t0 = ev[i0][0]
for s, e, n in ev[i0:packs[-1] + 40]:
    print('%9.1f us  +%7.1f us  %s' % ((s - t0)/1e3, (e - s)/1e3, n))
PY
