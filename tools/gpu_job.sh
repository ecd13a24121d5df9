#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): a sequence of steps, each under its own timeout, logging to gpurun_out/$TAG/.
# A step that is killed by its timeout (or by a signal) ends the job: no further GPU step is started after a hang.
#   usage: tools/gpu_job.sh TAG "name|timeout_s|command" ...
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; tmo=${rest%%|*}; cmd=${rest#*|}
  echo "== $name (timeout ${tmo}s): $cmd"
  start=$(date +%s)
  timeout -k 10 $tmo bash -c "$cmd" > $OUT/$name.log 2>&1
  rc=$?
  echo "== $name rc=$rc in $(( $(date +%s) - start ))s"; tail -n 6 $OUT/$name.log
  if [ $rc -ge 124 ]; then echo "== $name was killed (rc $rc): stopping the job"; exit $rc; fi
done
exit 0
