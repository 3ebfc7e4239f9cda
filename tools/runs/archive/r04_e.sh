# round 4, call e: core ranks above 64 on the projector route; LDS-tiled product in the product path
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
CFG5="-model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32"
tools/gpu_steps.sh \
 "r04e_tests_big|600|python -m pytest tests/test_gpu_tucker.py -x -q -k 'above_64' --durations=5" \
 "r04e_tests_tucker|600|python -m pytest tests/test_gpu_tucker.py -x -q --durations=5" \
 "r04e_nsprod_400|120|tools/nsprod_bench 400" \
 "r04e_cfg5|100|$B/test_ALS $CFG5 -filename gpurun_out/r04e_cfg5.csv" \
 "r04e_tests_full|600|python -m pytest tests/test_gpu_fullsize.py -x -q -k 'cfg5 or order6'"
