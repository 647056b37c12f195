#!/bin/bash
# usage: tools/gpurun_retry.sh TIMEOUT 'command'   -- gpurun, retried every 45 s while the pod's GPU slots are busy (exit code 3)
for i in $(seq 1 60); do
    /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
    rc=$?
    [ $rc -ne 3 ] && exit $rc
    sleep 45
done
exit 3
