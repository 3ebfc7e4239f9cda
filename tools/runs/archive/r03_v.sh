cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B2="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-config-records"
S=""
for i in 1 2; do
for c in "40:0" "16:0" "8:0" "24:0" "16:64" "8:64" "12:64" "6:64"; do
  pm=${c%%:*}; lds=${c##*:}
  S="$S \"r03v_p${pm}_l${lds}_$i|100|PPALS_PERSIST_MULT=$pm PPALS_SCAN_LDS_KB=$lds $B2\""
done; done
eval tools/gpu_steps.sh $S
