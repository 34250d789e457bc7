#!/bin/bash
# usage: scratch/gpu_retry.sh <timeout_s> '<command>'  -- gpurun, retried while the pod has no free GPU slot (exit code 3: nothing charged)
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@" > /tmp/gpu_retry.out 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then tail -30 /tmp/gpu_retry.out; exit $rc; fi
  sleep 100
done
echo "no slot after 40 tries"; exit 3
