#!/bin/bash
# Repeats the C++ jerkcar host (tests/cpp/jerkcar_host.cpp) until one run dies, line-buffered so that the number of rows written before
# the death is known, with core dumps on; a core is opened with rocgdb for the backtrace.  usage: scripts/flake_hunt.sh <kind> <runs>
kind=${1:-information}; runs=${2:-300}
G=tests/golden/jerkcar
ulimit -c unlimited
cd /tmp
fails=0
for i in $(seq 1 $runs); do
  stdbuf -oL /tmp/gokalman_amd_jerkcar_host $kind $OLDPWD/$G/uvec.csv $OLDPWD/$G/yacchist.csv $OLDPWD/$G/yposhist.csv > /tmp/fh.out 2> /tmp/fh.err
  rc=$?
  if [ $rc -ne 0 ]; then
    fails=$((fails+1))
    echo "run $i rc=$rc rows=$(wc -l < /tmp/fh.out)"; tail -c 3000 /tmp/fh.err
    core=""
    if [ -n "$core" ]; then /opt/rocm/bin/rocgdb -batch -ex "bt 25" -ex "info sharedlibrary" /tmp/gokalman_amd_jerkcar_host $core 2>&1 | grep -v "^\[New\|^warning" | head -60; rm -f /tmp/core*; fi
  fi
done
echo "$kind: $fails failures / $runs runs"
