# PMC passes for the conv-stack lab binary:  gpurun -- 'bash tools/experiments/cs_pmc.sh 128 fwd 2'
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/tools/experiments/cs_lab.bin
rocprofv3 -L 2>/dev/null | grep -io "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INSTS_VMEM[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_WAIT_INST_LDS\|SQ_INSTS_SMEM[A-Z_]*" | sort -u | tr '\n' ' '; echo
rm -rf /tmp/c1 /tmp/c2 /tmp/c3
rocprofv3 --kernel-trace -f csv -d /tmp/c1 -o p --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU -- $B "$@" > /tmp/c1.log 2>&1
rocprofv3 --kernel-trace -f csv -d /tmp/c2 -o p --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS -- $B "$@" > /tmp/c2.log 2>&1
rocprofv3 --kernel-trace -f csv -d /tmp/c3 -o p --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH -- $B "$@" > /tmp/c3.log 2>&1
for d in c1 c2 c3; do python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find /tmp/$d -name '*counter_collection.csv' | head -1) | grep -A12 "k_conv_stack" | head -14; done
