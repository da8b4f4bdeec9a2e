#!/bin/bash
# samples rocm-smi clocks/power while a command runs:  tools/smi_watch.sh <logfile> <cmd...>
LOG=$1; shift
( while true; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "sclk|Power|Temperature \(Sensor junction" | tr '\n' ' ' ; echo; sleep 0.15; done ) > $LOG &
WPID=$!
"$@"
kill $WPID
