# round-2 evidence: per-kernel times of the default bench under rocprofv3 --kernel-trace --stats, then PMC passes
# (one counter group per pass, never together with a trace) over tools/pmc_workload.py, then the phase stamps.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || echo "FAILED stats" >> $O/fail.txt
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
  D=$O/pmc_$(echo $SET | cut -c1-14 | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $SET --output-format csv -d $D -- python3 tools/pmc_workload.py 1000 3 > $D.log 2>&1 || echo "FAILED $SET" >> $O/fail.txt
done
python3 tools/pmc_summary.py $O > $O/pmc_summary.txt 2>&1
I8IE_PCONV_STAMPS=1 timeout -k 10 100 python3 tools/bench_layer.py 51 1 conv2,conv3,conv4,conv5 2>&1 | grep stamps | sort | uniq -c | sort -rn | head -12 > $O/pconv_stamps.txt
timeout -k 10 100 python3 tools/bench_layer.py 11,0,70 20 conv2,conv3,conv4,conv5 > $O/bench_layer.txt 2>&1
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/stats/*/*kernel_trace.csv 2>/dev/null
ls -la $O
tail -3 $O/pmc_summary.txt
