#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 120 ./tools/plane_layout > $OUT/r2_plane_layout.log 2>&1; echo rc=$?; cat $OUT/r2_plane_layout.log
