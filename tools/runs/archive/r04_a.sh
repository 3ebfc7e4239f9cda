# round 4, first call: state of the tree on this box + a kernel TRACE of the cfg5 run (timeline of one
# HOOI sweep: where the gaps are).  usage: tools/runs/r04_a.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
tools/gpu_steps.sh \
 "r04a_bench|500|python bench.py --gpus 1 --steps 20 --warmup 5" \
 "r04a_trace_cfg5|300|rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04a_trace_cfg5 -o r04a -- $B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r04a_cfg5_tucker_prof.csv" \
 "r04a_cfg5|200|PPALS_EIG_DEBUG=1 $B/test_ALS -model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32 -filename gpurun_out/r04a_cfg5_tucker.csv" \
 "r04a_nsprod|100|tools/nsprod_bench 400 && tools/nsprod_bench 1344" \
 "r04a_tests|900|python -m pytest tests -m gpu -x -q --durations=8"
