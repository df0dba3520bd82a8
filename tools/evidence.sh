# Round evidence: GPU tests, bench under rocprofv3 --kernel-trace --stats, the two PMC passes, SQ counters, phase split, plain bench.
# Run with gpurun (bash tools/evidence.sh); tools/collect_profiles.sh copies the summaries into profiles/round3.
set -e
cd /root/repo
export TMPDIR=/tmp
E=gpurun_out/ev3
mkdir -p $E
echo "== gpu tests"; timeout -k 10 900 python -m pytest tests -x -q -m gpu > $E/pytest_gpu.log 2>&1; tail -3 $E/pytest_gpu.log
echo "== pmc fetch"; cd /tmp; timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d /root/repo/$E/pmc_fetch -o f --output-format csv -- python /root/repo/tools/pmc_traffic.py > /root/repo/$E/pmc_fetch.log 2>&1
echo "== pmc write"; timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d /root/repo/$E/pmc_write -o w --output-format csv -- python /root/repo/tools/pmc_traffic.py > /root/repo/$E/pmc_write.log 2>&1
cd /root/repo; mkdir -p profiles/round3
python tools/pmc_summarize.py $E/pmc_fetch $E/pmc_write profiles/round3/traffic.json > $E/traffic_summary.txt 2>&1; cp profiles/round3/traffic.json $E/traffic.json; tail -30 $E/traffic_summary.txt
echo "== sq counters"; cd /tmp; timeout -k 10 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS -d /root/repo/$E/pmc_sq -o q --output-format csv -- python /root/repo/tools/perf_covis.py --sessions 14571582 --reps 1 > /root/repo/$E/pmc_sq.log 2>&1
cd /root/repo; python tools/pmc_sq_summarize.py $E/pmc_sq > $E/pmc_sq_summary.txt 2>&1
echo "== instruction counts"; cd /tmp; timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SMEM -d /root/repo/$E/pmc_inst -o q --output-format csv -- python /root/repo/tools/perf_covis.py --sessions 14571582 --reps 1 > /root/repo/$E/pmc_inst.log 2>&1
cd /root/repo
echo "== phase profile (make prof build)"; test -f otto-multi-objective-recommender-system_amd/csrc/libotto_amd_prof.so && timeout -k 10 300 python tools/perf_covis.py --sessions 14571582 --reps 1 --prof > $E/phase_split.log 2>&1
echo "== K1 debug skips"; for sk in 16 32 64 128 240; do timeout -k 10 100 python tools/perf_covis.py --sessions 14571582 --reps 2 --skip $sk 2>&1 | grep timings | tail -n 1 | sed "s/^/debug_skip $sk /" >> $E/k1_debug_skip.log; done
echo "== candidate lookup: counters, phase split, debug skips"; cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS -d /root/repo/$E/cand_sq -o q --output-format csv -- python /root/repo/tools/perf_cand.py --reps 1 > /root/repo/$E/cand_sq.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SMEM -d /root/repo/$E/cand_inst -o q --output-format csv -- python /root/repo/tools/perf_cand.py --reps 1 > /root/repo/$E/cand_inst.log 2>&1
cd /root/repo
(python tools/pmc_sq_summarize.py $E/cand_sq | grep -i "ratios\|cand\|recency"; python tools/pmc_inst_summarize.py $E/cand_inst | grep -i "cand\|recency") > $E/cand_pmc_summary.txt 2>&1
timeout -k 10 300 python tools/perf_cand.py --reps 2 > $E/cand_perf.log 2>&1; grep -v "rep 0" $E/cand_perf.log | tail -n 8
: > $E/cand_phase_split.log
for d in 0 1; do echo "OTTO_CAND_DEBUG=$d (1: no inserts -- the tables stay empty, so the selection is trivial too)" >> $E/cand_phase_split.log; OTTO_CAND_DEBUG=$d OTTO_AMD_LIB=$PWD/otto-multi-objective-recommender-system_amd/csrc/libotto_amd_prof.so timeout -k 10 300 python tools/perf_cand.py --reps 1 2>&1 | grep "phase-prof. k_cand" | head -n 2 >> $E/cand_phase_split.log; done
echo "== next rows"; timeout -k 10 400 python tools/perf_next_rows.py > $E/next_rows_perf.log 2>&1; tail -5 $E/next_rows_perf.log
echo "== bench under rocprof"; cd /tmp; timeout -k 10 700 rocprofv3 --kernel-trace --stats -d /root/repo/$E/prof -o r1 --output-format csv -- python /root/repo/bench.py > /root/repo/$E/bench_rocprof.log 2>&1; cd /root/repo; grep '"metric"' $E/bench_rocprof.log | cut -c1-300
echo "== plain bench"; timeout -k 10 600 python bench.py > $E/bench_plain.log 2>&1; grep '"metric"' $E/bench_plain.log | cut -c1-300
