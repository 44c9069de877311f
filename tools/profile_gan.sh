#!/bin/bash
# Evidence for BASELINE config #4 (tools/forth/gan_steps.4th = examples/t4_40b.4th nets, N = 256):  gpurun -- 'bash tools/profile_gan.sh r04'
# three separate runs of the VM: kernel trace, then two --pmc passes (counters never together with sys/hip/hsa traces)
set -u
TAG=${1:-r0x}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
S=$R/${2:-tools/forth/gan_steps.4th}
RAW=/tmp/t4prof_$TAG            # raw rocprofv3 output (databases, per-dispatch CSVs: tens of MB) stays on the box; only the summaries go to gpurun_out/
mkdir -p "$O" "$RAW"
cd /tmp && export TMPDIR=/tmp
T=$R/tensorforth_amd/ten4
$T < $S | grep -i "ms_for" > "$O/gan_plain.txt"
timeout 600 rocprofv3 --kernel-trace --stats -d "$RAW/gan_kt" -o gan -- $T < $S > "$O/gan_kt.log" 2>&1
timeout ${PMC_TIMEOUT:-600} rocprofv3 --kernel-trace -f csv -d "$RAW/gan_pmc_mfma" -o gan --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -- $T < $S > "$O/gan_pmc_mfma.log" 2>&1
timeout ${PMC_TIMEOUT:-600} rocprofv3 --kernel-trace -f csv -d "$RAW/gan_pmc_hbm" -o gan --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -- $T < $S > "$O/gan_pmc_hbm.log" 2>&1
{ cat "$O/gan_plain.txt"; python "$R/tools/rocpd_summary.py" "$(find "$RAW/gan_kt" -name '*.db' | head -1)"; } > "$O/gan_kernel_trace.txt"
for p in mfma hbm; do
  python "$R/tools/pmc_summary.py" "$(find "$RAW/gan_pmc_$p" -name '*counter_collection.csv' | head -1)" > "$O/gan_pmc_$p.txt"
done
head -24 "$O/gan_kernel_trace.txt" | cut -c1-76,112-170
