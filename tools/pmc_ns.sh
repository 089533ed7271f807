# PMC counters of the n=512 NS GEMM launches (two passes).  usage: bash tools/pmc_ns.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/ns1.py <<PY
import ctypes, sys
sys.path.insert(0, '$R/style-transfer-pytorch_amd')
from style_transfer import _hip
lib = _hip.load_library()
f, b = ctypes.c_double(), ctypes.c_double()
_hip._check(lib.st_op_sqrtm_time(512, 2, ctypes.byref(f), ctypes.byref(b), None))
print(f.value, b.value)
PY
rm -rf $R/gpurun_out/pmcns; mkdir -p $R/gpurun_out/pmcns
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d $R/gpurun_out/pmcns/a -o a --output-format csv -- python /tmp/ns1.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TA_DATA_STALL_CYCLES_sum -d $R/gpurun_out/pmcns/b -o b --output-format csv -- python /tmp/ns1.py > $R/gpurun_out/pmcns/b.log 2>&1
ls $R/gpurun_out/pmcns/a $R/gpurun_out/pmcns/b; tail -3 $R/gpurun_out/pmcns/b.log
# third pass (round 5): LDS and matrix-pipe counters
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS -d $R/gpurun_out/pmcns/c -o c --output-format csv -- python /tmp/ns1.py > $R/gpurun_out/pmcns/c.log 2>&1
tail -2 $R/gpurun_out/pmcns/c.log
python $R/tools/pmc_ns_summary.py $R/gpurun_out/pmcns > $R/gpurun_out/pmcns/summary.md 2>&1; cat $R/gpurun_out/pmcns/summary.md
