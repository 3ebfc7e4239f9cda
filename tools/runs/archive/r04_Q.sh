# Tucker order 3: multi-sweep dimension tree (3 scans per 2 HOOI sweeps)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
CFG5="-model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32"
tools/gpu_steps.sh \
 "r04Q_tucker_tests|1000|python -m pytest tests/test_gpu_tucker.py tests/test_gpu_driver.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz_campaign.py -m gpu -x -q -k 'tucker or Tucker or cfg5 or hooi'" \
 "r04Q_cfg5|200|$B/test_ALS $CFG5 -filename gpurun_out/r04Q_cfg5_tucker.csv" \
 "r04Q_cfg5_tree|200|PPALS_TUCKER_CHAIN=tree $B/test_ALS $CFG5 -filename gpurun_out/r04Q_cfg5_tucker_tree.csv" \
 "r04Q_cfg5_b|200|$B/test_ALS $CFG5 -filename gpurun_out/r04Q_cfg5_tucker_b.csv" \
 "r04Q_cfg5_tree_b|200|PPALS_TUCKER_CHAIN=tree $B/test_ALS $CFG5 -filename gpurun_out/r04Q_cfg5_tucker_tree_b.csv" \
 "r04Q_cfg5_log|200|PPALS_EIG_DEBUG=1 $B/test_ALS $CFG5 -filename gpurun_out/r04Q_cfg5_tucker_log.csv" \
 "r04Q_prof_cfg5|300|rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04Q_prof_cfg5 -o r04Q -- $B/test_ALS $CFG5 -filename gpurun_out/r04Q_cfg5_tucker_prof.csv"
