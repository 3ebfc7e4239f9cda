cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-config-records"
A="PPALS_DEBUG_ADDR=1 PPALS_PLACE_BLOCKS=1 PPALS_PLACE_LAYOUTS=1 PPALS_PLACE_STORE_KIND=0 PPALS_SCAN_NT_MB=100000000"
N="PPALS_DEBUG_ADDR=1"
S=""
for i in 1 2 3 4; do
  S="$S \"r03s_A$i|100|$A $B\" \"r03s_N$i|100|$N $B\""
done
eval tools/gpu_steps.sh $S
