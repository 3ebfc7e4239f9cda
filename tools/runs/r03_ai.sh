cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
C="$B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename"
tools/gpu_steps.sh \
 "r03ai_c1|100|$C gpurun_out/r03ai_c1.csv" "r03ai_c2|100|$C gpurun_out/r03ai_c2.csv" \
 "r03ai_prof|300|rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03ai_prof -o r03ai -- $C gpurun_out/r03ai_prof.csv"
