# round 5, call q: the HOSVD SYRK without the sub-tiles past the edge / below the diagonal — Tucker tests, then cfg5
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
RP="rocprofv3 --kernel-trace --stats --output-format csv"
CFG5="-model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32"
tools/gpu_steps.sh \
 "r05q_tests|1100|python -m pytest tests/test_gpu_tucker.py tests/test_gpu_fullsize.py tests/test_golden.py -m gpu -x -q" \
 "r05q_prof_cfg5|300|$RP -d gpurun_out/r05q_prof_cfg5 -o p -- $B/test_ALS $CFG5 -filename gpurun_out/r05q_cfg5.csv" 
f=$(find gpurun_out/r05q_prof_cfg5 -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r05q_cfg5_kernel_stats.csv; rm -rf gpurun_out/r05q_prof_cfg5
grep "syrk\|unfold" gpurun_out/r05q_cfg5_kernel_stats.csv | cut -c1-200
