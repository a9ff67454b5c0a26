#!/bin/bash
# tools/queue_map_probe.py over explicit replica -> creation-slot maps
run() { env "$@" python tools/queue_map_probe.py 2>&1 | tail -1; }
for m in ${MAPS:-0,1,0,1,0,1 0,1,0,1 0,1,2,0,1,0,1 0,1,2,3,0,1,0,1 0,1,2,3,0,1,2 0,1,0,1,0,1,0,1 0,0,0 0,1,2 0,1,3 0,1,0,1,2,3 0,1,2,0,1,2,0,1 0,1,0,1,0,1,2,3 0,1,2,3,0,1,2,0,1,0,1}; do run QMAP=$m; done
