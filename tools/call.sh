#!/bin/bash
# builds, then one gpurun call (retried while the pod's GPU slots are busy: nothing is charged then); prints the call's tail
# usage: tools/call.sh '<command on the box>'
cd "$(dirname "$0")/.." || exit 1
tools/build.sh || exit 1
for i in 1 2 3 4 5 6 7 8; do
  timeout 2400 /usr/local/graft/bin/gpurun --timeout ${GPU_TIMEOUT:-900} -- "$1" > /tmp/call.log 2>&1
  if grep -q "status=transient" /tmp/call.log; then sleep 60; else break; fi
done
tail -${TAIL:-9} /tmp/call.log
