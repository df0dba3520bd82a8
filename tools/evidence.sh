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
echo "== lds atomics ubench"; (test -x tools/ubench/lds_atomics || hipcc --offload-arch=gfx950 -O3 -o tools/ubench/lds_atomics tools/ubench/lds_atomics.hip) && timeout -k 10 120 tools/ubench/lds_atomics > gpurun_out/ev2/lds_atomics.txt 2>&1
echo "== plain bench"; timeout -k 10 400 python bench.py > gpurun_out/ev2/bench_plain.log 2>&1; grep '"metric"' gpurun_out/ev2/bench_plain.log | cut -c1-200
