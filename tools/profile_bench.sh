#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box:  gpurun -- 'bash tools/profile_bench.sh r1e'
# (four separate runs of bench.py: plain, kernel trace, and three --pmc passes; counters never together with sys/hip/hsa traces)
set -u
TAG=${1:-r1x}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
RAW=/tmp/t4prof_$TAG            # raw rocprofv3 output (databases, per-dispatch CSVs: tens of MB) stays on the box; only the summaries go to gpurun_out/
mkdir -p "$O" "$RAW"
cd /tmp && export TMPDIR=/tmp
python "$R/bench.py" > "$O/bench.json" 2> "$O/bench.err"
B="python $R/bench.py --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --kernel-trace --stats -d "$RAW/kt" -o bench -- $B > "$O/kt.log" 2>&1
timeout 600 rocprofv3 --kernel-trace -f csv -d "$RAW/pmc_mfma" -o bench --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -- $B > "$O/pmc_mfma.log" 2>&1
timeout 600 rocprofv3 --kernel-trace -f csv -d "$RAW/pmc_hbm" -o bench --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -- $B > "$O/pmc_hbm.log" 2>&1
timeout 600 rocprofv3 --kernel-trace -f csv -d "$RAW/pmc_lds" -o bench --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS -- $B > "$O/pmc_lds.log" 2>&1
DB=$(find "$RAW/kt" -name '*.db' | head -1)
python "$R/tools/rocpd_summary.py" "$DB" > "$O/kernel_trace.txt"
for p in mfma hbm lds; do
  C=$(find "$RAW/pmc_$p" -name '*counter_collection.csv' | head -1)
  python "$R/tools/pmc_summary.py" "$C" > "$O/pmc_$p.txt"
done
python "$R/tools/traffic_json.py" "$(find "$RAW/pmc_hbm" -name '*counter_collection.csv' | head -1)" "$O/bench_traffic.json" "TCC_EA0 request counters, separate --pmc pass of bench.py, tools/profile_bench.sh $TAG"
tail -1 "$O/bench.json"
head -25 "$O/kernel_trace.txt"
