R=$GRAFT_REPO_ROOT
python bench.py --math fp32 --no-c3 --no-cpu-baseline --no-dataset --steps 3 --warmup 1 > $R/gpurun_out/r03_bench_q_auto.json 2> $R/gpurun_out/r03_bench_q_auto.err; python -c "import json;d=json.load(open('$R/gpurun_out/r03_bench_q_auto.json'));print('auto ms/step',d['ms_per_step'])"
PCNN_SPEC_T=32 python bench.py --math fp32 --no-c3 --no-cpu-baseline --no-dataset --steps 3 --warmup 1 > $R/gpurun_out/r03_bench_q_t32.json 2> $R/gpurun_out/r03_bench_q_t32.err; python -c "import json;d=json.load(open('$R/gpurun_out/r03_bench_q_t32.json'));print('T=32 ms/step',d['ms_per_step'])"
python -m pytest tests -m gpu -x -v > $R/gpurun_out/r03_tests_c.log 2>&1; echo "tests rc=$?"; tail -3 $R/gpurun_out/r03_tests_c.log
