cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc -- python3 tools/bench_layer.py 50,70 2 conv2,conv3 > $O/pmc.log 2>&1
python3 tools/pmc_summary.py $O conv_kernel > $O/summary.txt 2>&1
cat $O/summary.txt
