cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1
mkdir -p $O
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/pmc1 -- python3 tools/bench_linear.py 0 3 1000 > $O/pmc1.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc2 -- python3 tools/bench_linear.py 0 3 1000 > $O/pmc2.log 2>&1
python3 tools/pmc_summary.py $O lgemm > $O/summary.txt 2>&1
cat $O/summary.txt
