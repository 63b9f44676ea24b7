bash tests/tools/prof_stats.sh r01h > gpurun_out/r01h_table.txt 2>&1; tail -22 gpurun_out/r01h_table.txt
bash tests/tools/prof_pmc.sh r01h > gpurun_out/r01h_pmc_table.txt 2>&1; tail -16 gpurun_out/r01h_pmc_table.txt
timeout 900 python bench.py > gpurun_out/r01h_bench_default.json 2> gpurun_out/r01h_bench_default.err; tail -c 1500 gpurun_out/r01h_bench_default.json
