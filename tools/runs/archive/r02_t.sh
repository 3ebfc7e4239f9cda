#!/bin/bash
# round 2, step t: padded resident layouts on the reference scripts' shapes (s = 50 order 6, 324^4)
B=pairwise-perturbation_amd/bin
RP="rocprofv3 --kernel-trace --stats --output-format csv"
S6="-model CP -tensor r -dim 6 -size 50 -rank 6 -pp 0 -maxiter 30 -prec 32"
P8="-model CP -tensor p -dim 8 -size 18 -rank 2 -pp 0 -maxiter 30 -prec 32"
exec tools/gpu_steps.sh \
 "r02t_tests|900|python -m pytest tests/test_gpu_padded.py tests/test_gpu_cp.py -m gpu -x -q" \
 "r02t_s6_pad|500|PPALS_DEBUG_ADDR=1 $B/test_ALS $S6 -filename gpurun_out/r02t_script_cp6_s50_r6_pad.csv" \
 "r02t_s6_nopad|500|PPALS_PAD_LAYOUT=0 $B/test_ALS $S6 -filename gpurun_out/r02t_script_cp6_s50_r6_nopad.csv" \
 "r02t_p8_pad|500|$B/test_ALS $P8 -filename gpurun_out/r02t_script_p8_pad.csv" \
 "r02t_p8_nopad|500|PPALS_PAD_LAYOUT=0 $B/test_ALS $P8 -filename gpurun_out/r02t_script_p8_nopad.csv" \
 "r02t_prof_s6|600|$RP -d gpurun_out/r02t_prof_s6 -o r02t -- $B/test_ALS $S6 -filename gpurun_out/r02t_script_cp6_s50_r6_prof.csv"
