# bench.py with HBM traffic counted in the run (child passes under rocprofv3 --pmc)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh \
 "r04D_bench|900|time python bench.py --gpus 1 --steps 20 --warmup 5"
