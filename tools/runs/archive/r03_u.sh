cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B4="python bench.py --workload cp4_s400_r20 --steps 4 --warmup 1 --no-cpu-baseline --no-config-records"
B2="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-config-records"
tools/gpu_steps.sh "r03u_A1|300|PPALS_SCAN_ACC2=0 $B4" "r03u_N1|300|$B4" "r03u_A2|300|PPALS_SCAN_ACC2=0 $B4" "r03u_N2|300|$B4" \
  "r03u_a1|100|PPALS_SCAN_ACC2=0 $B2" "r03u_n1|100|$B2" "r03u_a2|100|PPALS_SCAN_ACC2=0 $B2" "r03u_n2|100|$B2" "r03u_a3|100|PPALS_SCAN_ACC2=0 $B2" "r03u_n3|100|$B2"
