cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
tools/gpu_steps.sh \
 "r02a_bench|300|python bench.py --steps 20 --warmup 5" \
 "r02a_prof_bench|300|rocprofv3 --kernel-trace --stats -d gpurun_out/r02a_prof_bench -o r02a -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline" \
 "r02a_prof_pp1|300|rocprofv3 --kernel-trace --stats -d gpurun_out/r02a_prof_pp1 -o r02a -- $B/test_ALS -model CP -tensor r -dim 4 -size 200 -rank 10 -pp 1 -pp_res_tol 0.01 -maxiter 100 -filename gpurun_out/r02a_pp1.csv" \
 "r02a_prof_tucker|300|rocprofv3 --kernel-trace --stats -d gpurun_out/r02a_prof_tucker -o r02a -- $B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 20 -filename gpurun_out/r02a_tucker.csv" \
 "r02a_ppbench|200|$B/pp_bench -model CP -tensor r -dim 4 -size 200 -rank 10 -maxiter 5 -filename gpurun_out/r02a_pp_bench.csv"
