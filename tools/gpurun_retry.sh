#!/bin/bash
# run one gpurun call; when no slot / box is free right now (exit code 3: nothing charged) try again after two minutes
for i in 1 2 3 4 5 6 7 8; do
    /usr/local/graft/bin/gpurun "$@"
    rc=$?
    if [ $rc -ne 3 ]; then exit $rc; fi
    echo "[gpurun_retry] no slot (try $i), sleeping 120 s"
    sleep 120
done
exit 3
