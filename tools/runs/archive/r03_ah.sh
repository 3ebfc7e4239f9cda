cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
tools/gpu_steps.sh "r03ah_update_bench|120|timeout -k 5 100 tools/update_bench" \
 "r03ah_tests|1000|python -m pytest tests -m gpu -x -q" \
 "r03ah_ppbench|200|$B/pp_bench -model CP -tensor r -dim 4 -size 200 -rank 10 -maxiter 5 -prec 32 -filename gpurun_out/r03ah_pp_bench_cp.csv" \
 "r03ah_shard_probe|300|python tools/shard_probe.py 200 10 8 msdt"
