# round-4 evidence, one call: the default bench, per-kernel times under rocprofv3 --kernel-trace --stats, PMC passes (one counter
# group per pass, never together with a trace) over tools/pmc_workload.py, the per-GPU shard sizes, the `--gpus 2` rehearsal started
# without a launcher, layer timings of the convolution variants, phase stamps from the diagnostic build.
# usage: bash tools/r04_profile.sh <tag>
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/$1
mkdir -p $O
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || echo "FAILED bench default" >> $O/fail.txt
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || echo "FAILED stats" >> $O/fail.txt
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/stats
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
  D=$O/pmc_$(echo $SET | cut -c1-14 | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $SET --output-format csv -d $D -- python3 tools/pmc_workload.py 1000 3 > $D.log 2>&1 || echo "FAILED $SET" >> $O/fail.txt
done
python3 tools/pmc_summary.py $O > $O/pmc_summary.txt 2>&1
rm -rf $O/pmc_*/
python3 bench.py --batch 125 --steps 100 --warmup 5 --no-cpu-baseline --graph off > $O/bench_batch125_eager.json 2> $O/b125e.err || echo "FAILED b125 eager" >> $O/fail.txt
python3 bench.py --batch 125 --steps 100 --warmup 5 --no-cpu-baseline > $O/bench_batch125.json 2> $O/b125.err || echo "FAILED b125" >> $O/fail.txt
python3 bench.py --batch 125 --steps 100 --warmup 5 --no-cpu-baseline --force-dist > $O/bench_batch125_forcedist.json 2> $O/b125fd.err || echo "FAILED b125 forcedist" >> $O/fail.txt
for b in 250 500; do python3 bench.py --batch $b --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_batch$b.json 2> $O/b$b.err; done
I8IE_BENCH_REHEARSE=1 I8IE_BENCH_NO_PREWARM=1 python3 bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_gpus2_rehearsal.json 2> $O/bg2.err || echo "FAILED gpus 2 rehearsal" >> $O/fail.txt
timeout -k 10 100 python3 tools/bench_layer.py 11,0,70 20 conv2,conv3,conv4,conv5 > $O/bench_layer.txt 2>&1
timeout -k 10 100 python3 tools/bench_linear.py 0,11 30 1000,500 > $O/bench_linear.txt 2>&1
if [ -f tools/diag/libi8ie_hip_diag.so ]; then
  I8IE_LIB=tools/diag/libi8ie_hip_diag.so I8IE_STEM_STAMPS=1 timeout -k 10 200 python3 tools/bench_stem.py 3 1000 2>&1 | grep stem_stamps | sed -n 1p > $O/stem_stamps.txt
  I8IE_LIB=tools/diag/libi8ie_hip_diag.so I8IE_PCONV_STAMPS=1 timeout -k 10 100 python3 tools/bench_layer.py 51 1 conv2,conv3,conv4,conv5 2>&1 | grep stamps | sort | uniq -c | sort -rn | head -12 > $O/pconv_stamps.txt
  I8IE_LIB=tools/diag/libi8ie_hip_diag.so timeout -k 10 100 python3 tools/bench_layer.py 50,55 20 conv2,conv5 > $O/bench_layer_dconv.txt 2>&1
fi
ls -la $O
cat $O/fail.txt 2>/dev/null
tail -3 $O/pmc_summary.txt
