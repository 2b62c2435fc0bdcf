#!/bin/bash
# usage: bash profiles/gpujob.sh <tag> <timeout_s> <command...>   — retries while no slot / box is free (exit code 3), log in gpurun_out/<tag>.out
TAG=$1; TMO=$2; shift 2
mkdir -p gpurun_out
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $TMO -- "$@" > gpurun_out/$TAG.out 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then break; fi
  sleep 90
done
echo "gpujob exit $rc" >> gpurun_out/$TAG.out
