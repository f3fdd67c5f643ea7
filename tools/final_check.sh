#!/bin/bash
# Developer tool (GPU box): the end-of-round record -- GPU suite, smoke, the
# default bench line and the rocprofv3 kernel statistics of the same command.
#   gpurun --timeout 2700 -- 'bash tools/final_check.sh <tag>'
tag=${1:-final}
mkdir -p gpurun_out/prof_$tag
(time OPTY_CACHE_MANIFEST=$PWD/gpurun_out/${tag}_manifest.txt timeout 2400 python -m pytest tests -m gpu -q --timeout 900) > gpurun_out/${tag}_suite.log 2>&1
grep -a "passed\|failed" gpurun_out/${tag}_suite.log | tail -2
cp gpurun_out/parity_stats.json gpurun_out/${tag}_parity_stats.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
(time timeout 700 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err) 2>&1 | grep real
root=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -- \
    python $root/bench.py --no-cpu-baseline --no-extras > $root/gpurun_out/${tag}_bench_prof.json 2>/dev/null
cd $root
find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/${tag}_bench.json") if l.startswith("{")][-1])
c = d["config"]
print(d["value"], d["roofline"]["frac"], c["kernel_ms"], c["verify"]["ok"])
print(c["host_path_ms"])
print({k: (round(v.get("fused_hbm_frac", -1), 3), v.get("shard_1of8", {}).get("speedup_vs_whole"),
           v.get("host_path_us"), v.get("error")) for k, v in c["other_configs"].items()})
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
