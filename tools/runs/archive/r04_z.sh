# second stream released by a stored value (hipStreamWaitValue64) instead of an event marker
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
RP="rocprofv3 --kernel-trace --stats --output-format csv"
CFG5="-model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32"
tools/gpu_steps.sh \
 "r04z_tucker_tests|900|python -m pytest tests/test_gpu_tucker.py tests/test_gpu_driver.py -m gpu -x -q" \
 "r04z_cfg5|200|$B/test_ALS $CFG5 -filename gpurun_out/r04z_cfg5_tucker.csv" \
 "r04z_cfg5_event|200|PPALS_EIG_DEFER=2 $B/test_ALS $CFG5 -filename gpurun_out/r04z_cfg5_tucker_event.csv" \
 "r04z_cfg5_b|200|$B/test_ALS $CFG5 -filename gpurun_out/r04z_cfg5_tucker_b.csv" \
 "r04z_cfg5_event_b|200|PPALS_EIG_DEFER=2 $B/test_ALS $CFG5 -filename gpurun_out/r04z_cfg5_tucker_event_b.csv" \
 "r04z_prof_cfg5|300|$RP -d gpurun_out/r04z_prof_cfg5 -o r04z -- $B/test_ALS $CFG5 -filename gpurun_out/r04z_cfg5_tucker_prof.csv"
