# Round evidence: GPU tests, bench under rocprofv3 --kernel-trace --stats, the two PMC passes, plain bench. Run with gpurun; copy the results into profiles/.
set -e
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/ev2
echo "== gpu tests"; timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/ev2/pytest_gpu.log 2>&1; tail -3 gpurun_out/ev2/pytest_gpu.log
echo "== bench under rocprof"; cd /tmp; timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/ev2/prof -o r1 --output-format csv -- python /root/repo/bench.py > /root/repo/gpurun_out/ev2/bench_rocprof.log 2>&1; cd /root/repo; grep '"metric"' gpurun_out/ev2/bench_rocprof.log | cut -c1-300
echo "== pmc fetch"; cd /tmp; timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d /root/repo/gpurun_out/ev2/pmc_fetch -o f --output-format csv -- python /root/repo/tools/pmc_traffic.py > /root/repo/gpurun_out/ev2/pmc_fetch.log 2>&1
echo "== pmc write"; timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d /root/repo/gpurun_out/ev2/pmc_write -o w --output-format csv -- python /root/repo/tools/pmc_traffic.py > /root/repo/gpurun_out/ev2/pmc_write.log 2>&1
cd /root/repo; ls gpurun_out/ev2/pmc_fetch gpurun_out/ev2/pmc_write
python tools/pmc_summarize.py gpurun_out/ev2/pmc_fetch gpurun_out/ev2/pmc_write profiles/round2/traffic.json > gpurun_out/ev2/traffic_summary.txt 2>&1; cp profiles/round2/traffic.json gpurun_out/ev2/traffic.json
echo "== sq counters"; cd /tmp; timeout -k 10 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS -d /root/repo/gpurun_out/ev2/pmc_sq -o q --output-format csv -- python /root/repo/tools/perf_covis.py --sessions 14571582 --reps 1 > /root/repo/gpurun_out/ev2/pmc_sq.log 2>&1
cd /root/repo; python tools/pmc_sq_summarize.py gpurun_out/ev2/pmc_sq > gpurun_out/ev2/pmc_sq_summary.txt 2>&1
echo "== phase profile (make prof build)"; test -f otto-multi-objective-recommender-system_amd/csrc/libotto_amd_prof.so && timeout -k 10 300 python tools/perf_covis.py --sessions 14571582 --reps 1 --prof > gpurun_out/ev2/phase_split.log 2>&1
echo "== lds atomics ubench"; (test -x tools/ubench/lds_atomics || hipcc --offload-arch=gfx950 -O3 -o tools/ubench/lds_atomics tools/ubench/lds_atomics.hip) && timeout -k 10 120 tools/ubench/lds_atomics > gpurun_out/ev2/lds_atomics.txt 2>&1
(test -x tools/ubench/valu_rates || hipcc --offload-arch=gfx950 -O3 -w -o tools/ubench/valu_rates tools/ubench/valu_rates.hip) && timeout -k 10 120 tools/ubench/valu_rates > gpurun_out/ev2/valu_rates.txt 2>&1
echo "== next rows"; timeout -k 10 400 python tools/perf_next_rows.py > gpurun_out/ev2/next_rows_perf.log 2>&1; tail -5 gpurun_out/ev2/next_rows_perf.log
echo "== plain bench"; timeout -k 10 400 python bench.py > gpurun_out/ev2/bench_plain.log 2>&1; grep '"metric"' gpurun_out/ev2/bench_plain.log | cut -c1-200
