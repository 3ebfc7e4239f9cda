cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
S=""
for m in 16 4 8 32 64 128 16; do S="$S \"r03z_m$m|120|PPALS_PLACE_TUNE=0 PPALS_RANK_CHUNK_MULT=$m python tools/k10_probe.py\""; done
eval tools/gpu_steps.sh $S
