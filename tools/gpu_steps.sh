#!/bin/bash
# Run GPU steps in order on the GPU box; each step under its own timeout. A failing step (test
# assertion) does not stop the sequence, but a step that is KILLED (timeout / signal) does: no
# further GPU step is started after a hang.  usage: tools/gpu_steps.sh "name|seconds|command" ...
mkdir -p gpurun_out
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; secs="${rest%%|*}"; cmd="${rest#*|}"
  echo "=== step $name (limit ${secs}s): $cmd"
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== step $name exit=$rc"
  tail -n 12 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then
    echo "=== step $name was killed (rc=$rc): stopping, no further GPU steps"
    exit $rc
  fi
done
exit 0
