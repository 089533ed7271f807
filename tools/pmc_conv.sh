cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/conv_bench.py 512 3 | tail -9
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU -d $R/gpurun_out/pmcA -o a --output-format csv -- python $R/tools/conv_bench.py 512 3 conv4_2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_INSTS_MFMA -d $R/gpurun_out/pmcB -o b --output-format csv -- python $R/tools/conv_bench.py 512 3 conv4_2 > /dev/null 2>&1
ls $R/gpurun_out/pmcA $R/gpurun_out/pmcB
