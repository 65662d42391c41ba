cd $GRAFT_REPO_ROOT
bash tools/prof_all.sh r06 c3 c4 > gpurun_out/r6_prof_all.log 2>&1
python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
tail -c 200 gpurun_out/r06_bench.json
