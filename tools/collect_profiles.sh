# Copies the summaries of the last tools/evidence.sh run (gpurun_out/ev3, scratch) into profiles/round3 (tracked).
set -e
cd /root/repo
E=gpurun_out/ev3; P=profiles/round3
mkdir -p $P/pmc
cp $E/prof/r1_kernel_stats.csv $P/bench_kernel_stats.csv
cp $E/pmc_fetch/f_counter_collection.csv $P/pmc/fetch_counter_collection.csv
cp $E/pmc_write/w_counter_collection.csv $P/pmc/write_counter_collection.csv
cp $E/pmc_sq/q_counter_collection.csv $P/pmc/sq_counter_collection.csv
cp $E/pmc_inst/q_counter_collection.csv $P/pmc/inst_counter_collection.csv
cp $E/pmc_sq_summary.txt $P/pmc_sq_summary.txt
cp $E/traffic.json $P/traffic.json
cp $E/traffic_summary.txt $P/traffic_summary.txt
grep '"metric"' $E/bench_plain.log > $P/bench.json
grep '"metric"' $E/bench_rocprof.log > $P/bench_under_rocprof.json
cp $E/pytest_gpu.log $P/pytest_gpu.log
cp $E/next_rows_perf.log $P/next_rows_perf.log
cp $E/k1_debug_skip.log $P/k1_debug_skip.log
grep -v "amdgpu.ids" $E/phase_split.log > $P/phase_split_reduce.log
cp $E/cand_pmc_summary.txt $P/cand_pmc_summary.txt
cp $E/cand_phase_split.log $P/cand_phase_split.log
grep -v "rep 0" $E/cand_perf.log > $P/cand_perf.log
ls -la $P
