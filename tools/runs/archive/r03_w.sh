# rocprofv3 passes of the driver's command after the round's scan changes: kernel trace + stats, and
# the two PMC passes (separate runs, no trace domains beside them)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tools/gpu_steps.sh \
 "r03w_prof_bench|400|rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03w_prof_bench -o r03w -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-config-records" \
 "r03w_pmc_fetch|400|rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r03w_pmc_fetch -o r03w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config-records" \
 "r03w_pmc_write|400|rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r03w_pmc_write -o r03w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config-records"
