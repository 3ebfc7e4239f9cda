#!/bin/bash
# round 6, step j: new tests (strict preload, long-mode cold start, CLI flag space), coil-100 extents
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=r06j
tools/gpu_steps.sh \
 "${T}_tests|900|python -m pytest tests/test_gpu_tucker.py tests/test_gpu_driver.py -x -q -m gpu -k 'strict_preload or long_mode_cold or every_tensor_source or above_64 or flat_spectrum'" \
 "${T}_coil100|600|python bench.py --workload coil100" \
 "${T}_timelapse|600|python bench.py --workload timelapse"
tail -1 gpurun_out/${T}_coil100.log > gpurun_out/${T}_coil100.json
tail -1 gpurun_out/${T}_timelapse.log > gpurun_out/${T}_timelapse.json
