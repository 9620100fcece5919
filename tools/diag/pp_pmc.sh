set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02f_pmc
mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1 || true
for V in 0 21 22; do
 for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE"; do
  D=$O/v${V}_$(echo $SET | cut -c1-12 | tr ' ' '_')
  timeout -k 10 120 rocprofv3 --pmc $SET --output-format csv -d $D -- python3 tools/pp_probe.py conv5 $V 3 > $D.log 2>&1 || echo "FAILED $V $SET" >> $O/fail.txt
 done
done
python3 tools/pmc_summary.py $O pp_conv > $O/summary.txt 2>&1
cat $O/summary.txt | head -80
