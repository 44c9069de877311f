# PMC passes (MFMA busy / waits, LDS conflicts) for any Forth script, counters only with --kernel-trace:  gpurun -- 'bash tools/experiments/pmc_script.sh tools/forth/cifar_steps.4th'
cd /tmp && export TMPDIR=/tmp
S=$GRAFT_REPO_ROOT/$1
rm -rf /tmp/pm1 /tmp/pm2
rocprofv3 --kernel-trace -f csv -d /tmp/pm1 -o p --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -- $GRAFT_REPO_ROOT/tensorforth_amd/ten4 < $S > /tmp/pm1.log 2>&1
rocprofv3 --kernel-trace -f csv -d /tmp/pm2 -o p --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -- $GRAFT_REPO_ROOT/tensorforth_amd/ten4 < $S > /tmp/pm2.log 2>&1
for d in pm1 pm2; do python $GRAFT_REPO_ROOT/tools/pmc_summary.py $(find /tmp/$d -name '*counter_collection.csv' | head -1) | grep -A12 "${2:-k_convbig}" | head -${3:-80}; done
