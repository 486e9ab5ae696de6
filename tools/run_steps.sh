#!/bin/bash
# Runs the GPU steps given on stdin (one shell command per line, '#' comments allowed) one after the
# other, each under its own time limit (first word of the line, seconds), output appended to
# gpurun_out/<tag>.log; stops at the first step that fails or times out.
# usage: tools/run_steps.sh <tag> < tools/steps/verify.txt   (through gpurun: the steps file must be in the tree)
TAG=${1:?tag}
LOG=gpurun_out/$TAG.log
mkdir -p gpurun_out
: > $LOG
export TMPDIR=/tmp
while IFS= read -r line; do
  case "$line" in ''|'#'*) continue;; esac
  limit=${line%% *}; cmd=${line#* }
  echo "=== [$limit s] $cmd" | tee -a $LOG
  timeout -k 10 $limit bash -c "$cmd" >> $LOG 2>&1
  rc=$?
  echo "=== rc $rc" | tee -a $LOG
  if [ $rc -ne 0 ]; then echo "stopped at: $cmd"; tail -30 $LOG; exit $rc; fi
done
tail -5 $LOG
