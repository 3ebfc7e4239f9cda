cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-config-records"
A="PPALS_PLACE_BLOCKS=1 PPALS_PLACE_LAYOUTS=1"
C="PPALS_PLACE_BLOCKS=6 PPALS_PLACE_LAYOUTS=1 PPALS_PLACE_PER_ROOT=1"
D="PPALS_PLACE_BLOCKS=6 PPALS_PLACE_LAYOUTS=1"
E="PPALS_PLACE_BLOCKS=6 PPALS_PLACE_LAYOUTS=5"
F="PPALS_PLACE_BLOCKS=6 PPALS_PLACE_LAYOUTS=5 PPALS_PLACE_PER_ROOT=1"
S=""
for i in 1 2 3; do
  S="$S \"r03r_A$i|100|$A $B\" \"r03r_C$i|100|$C $B\" \"r03r_D$i|100|$D $B\" \"r03r_E$i|100|$E $B\" \"r03r_F$i|100|$F $B\""
done
eval tools/gpu_steps.sh $S
