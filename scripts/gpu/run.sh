#!/bin/bash
# run.sh <tag> <gpurun timeout s> <command...>: gpurun with retries while no box / slot is free (exit code 3: nothing was charged)
tag=$1; to=$2; shift 2
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $to -- "$@" > gpurun_out/${tag}_call.log 2>&1; rc=$?
  [ $rc -eq 3 ] || break
  sleep 90
done
echo "gpurun rc=$rc" >> gpurun_out/${tag}_call.log
