# round 5, call f: the skinny-shape contractions (k_mttv_s, j-split k_mttv_l, k_gram) — GPU suite, then the
# coil-100 / time-lapse lines and the kernel split of their exact sweeps and Tucker runs on image-like data
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
RP="rocprofv3 --kernel-trace --stats --output-format csv"
tools/gpu_steps.sh \
 "r05f_tests|1100|python -m pytest tests -m gpu -x -q --durations=8" \
 "r05f_coil100|400|PPALS_PLACE_MIN_MB=1000 python bench.py --workload coil100 --steps 20 --warmup 3" \
 "r05f_timelapse|400|PPALS_PLACE_MIN_MB=1000 python bench.py --workload timelapse --steps 20 --warmup 3" \
 "r05f_prof_coil|300|$RP -d gpurun_out/r05f_prof_coil -o p -- python3 bench.py --workload coil100 --steps 30 --warmup 3 --no-config-records" \
 "r05f_prof_tk_coil|300|$RP -d gpurun_out/r05f_prof_tk_coil -o p -- python3 tools/runs/real_tucker_probe.py coil100" \
 "r05f_prof_tk_tl|300|$RP -d gpurun_out/r05f_prof_tk_tl -o p -- python3 tools/runs/real_tucker_probe.py timelapse"
for n in coil100 timelapse; do tail -1 gpurun_out/r05f_$n.log > gpurun_out/r05f_$n.json; done
for d in prof_coil prof_tk_coil prof_tk_tl; do
  f=$(find gpurun_out/r05f_$d -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r05f_${d}_kernel_stats.csv; rm -rf gpurun_out/r05f_$d
done
