#!/bin/bash
# usage: tools/gp.sh <logfile> <timeout> '<command>'  -- starts a gpurun call in the background and returns once the box has the
# snapshot of the tree (the call shows as in flight for > 20 s), so that the tree can be edited again without racing the push.
log=$1; to=$2; cmd=$3
(gpurun --timeout $to -- "$cmd" > $log 2>&1 &)
for i in $(seq 1 120); do
  sleep 5
  el=$(gpurun --status 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); c=d.get('in_flight_call') or {}; print(c.get('elapsed_s', -1))" 2>/dev/null)
  if grep -q "status=" $log 2>/dev/null; then echo "finished already"; exit 0; fi
  if python3 -c "import sys; sys.exit(0 if float('$el' or -1) > 20 else 1)" 2>/dev/null; then echo "snapshot taken (elapsed $el s)"; exit 0; fi
done
echo "gave up waiting"
