#!/bin/bash
# Runs the given commands one after the other on the GPU box, each under its own `timeout -k 10 <seconds>`; a step that is killed by
# its limit (rc 124 / 137) ends the sequence -- no further GPU step is started behind a hang.  A step that merely FAILS (rc 1: a
# red test) does not.     usage: tools/gpu_seq.sh "<seconds> <command>" "<seconds> <command>" ...
for spec in "$@"; do
  secs=${spec%% *}; cmd=${spec#* }
  echo "[gpu_seq] ($secs s) $cmd"
  timeout -k 10 "$secs" bash -c "$cmd"; rc=$?
  echo "[gpu_seq] rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[gpu_seq] step hit its limit: stopping"; exit $rc; fi
done
