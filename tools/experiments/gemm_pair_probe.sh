cd $GRAFT_REPO_ROOT
for v in 4 12; do for shp in 1024x1024x256 1024x1024x512 1024x1024x1024 1024x1024x2048 512x512x1024; do T4K_GEMM_VARIANT=$v python tools/gemm_tune.py one $shp 2>&1 | grep variant; done; done
cd /tmp; export TMPDIR=/tmp
T4K_GEMM_VARIANT=12 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pair_pmc -o g -- python $GRAFT_REPO_ROOT/tools/gemm_tune.py one 1024x1024x1024 > /dev/null 2>&1
T4K_GEMM_VARIANT=4 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/nopair_pmc -o g -- python $GRAFT_REPO_ROOT/tools/gemm_tune.py one 1024x1024x1024 > /dev/null 2>&1
