#!/bin/bash
# kernel time + SQ counters of the many-channel dF kernel:  gpurun -- 'bash tools/experiments/conv_df_pmc.sh 256 32 64 64'
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/conv_df_pmc
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$O/kt" -o c -- python "$R/tools/experiments/conv_df_one.py" "$@" 2000 > "$O/kt.log" 2>&1
python "$R/tools/rocpd_summary.py" "$(find "$O/kt" -name '*.db' | head -1)" | grep -E "convbig|fold|colsum" | cut -c1-60,112-170
rm -rf "$O/kt"
timeout 300 rocprofv3 --kernel-trace -f csv -d "$O/p1" -o c --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -- python "$R/tools/experiments/conv_df_one.py" "$@" 40 > "$O/p1.log" 2>&1
python "$R/tools/pmc_summary.py" "$(find "$O/p1" -name '*counter_collection.csv' | head -1)" convbig_df
timeout 300 rocprofv3 --kernel-trace -f csv -d "$O/p2" -o c --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python "$R/tools/experiments/conv_df_one.py" "$@" 40 > "$O/p2.log" 2>&1
python "$R/tools/pmc_summary.py" "$(find "$O/p2" -name '*counter_collection.csv' | head -1)" convbig_df
rm -rf "$O/p1" "$O/p2"
