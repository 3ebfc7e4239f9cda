# round 5, call j: is the tail mode of the two-tile scan engaged at cfg5, and what does it buy? kernel stats of the
# cfg5 driver run with and without it (PPALS_SCAN_TAIL=0), and the [dtime] of both, same box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=pairwise-perturbation_amd/bin
RP="rocprofv3 --kernel-trace --stats --output-format csv"
CFG5="-model Tucker -tensor r2 -dim 3 -size 400 -rank 20 -pp 0 -maxiter 40 -prec 32"
tools/gpu_steps.sh \
 "r05j_prof_cfg5|300|$RP -d gpurun_out/r05j_prof_cfg5 -o p -- $B/test_ALS $CFG5 -filename gpurun_out/r05j_cfg5_prof.csv" \
 "r05j_cfg5_a|200|$B/test_ALS $CFG5 -filename gpurun_out/r05j_cfg5_tail.csv" \
 "r05j_cfg5_b|200|PPALS_SCAN_TAIL=0 $B/test_ALS $CFG5 -filename gpurun_out/r05j_cfg5_notail.csv" \
 "r05j_cfg5_c|200|$B/test_ALS $CFG5 -filename gpurun_out/r05j_cfg5_tail2.csv" \
 "r05j_cfg5_d|200|PPALS_SCAN_TAIL=0 $B/test_ALS $CFG5 -filename gpurun_out/r05j_cfg5_notail2.csv"
f=$(find gpurun_out/r05j_prof_cfg5 -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r05j_cfg5_kernel_stats.csv; rm -rf gpurun_out/r05j_prof_cfg5
for f in tail notail tail2 notail2; do echo $f; tail -2 gpurun_out/r05j_cfg5_$f.csv; done
grep "k_scan" gpurun_out/r05j_cfg5_kernel_stats.csv | cut -c1-200
