cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3c
mkdir -p $O
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/pmc1 -- python3 tools/bench_stem.py 2 1000 > $O/pmc1.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc2 -- python3 tools/bench_stem.py 2 1000 > $O/pmc2.log 2>&1
python3 tools/pmc_summary.py $O stem_conv > $O/summary.txt 2>&1
cat $O/summary.txt
