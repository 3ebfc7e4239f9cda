cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
tools/gpu_steps.sh \
 "r02j_tests|900|python -m pytest tests -m gpu -x -q" \
 "r02j_bench|420|python bench.py --steps 20 --warmup 5" \
 "r02j_prof_ppbench|300|rocprofv3 --kernel-trace --stats -d gpurun_out/r02j_prof_ppbench -o r02j -- $B/pp_bench -model CP -tensor r -dim 4 -size 200 -rank 10 -maxiter 5 -prec 32 -filename gpurun_out/r02j_pp_bench.csv"
