#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration per access pattern (measurement hygiene): builds tools/pmc_calib.hip on the box, two PMC passes
TAG=${TAG:-r03}
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/pmc_calib.hip -o /tmp/pmc_calib || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  O=$R/gpurun_out/calib_${TAG}_$c; rm -rf $O; mkdir -p $O
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O -o pmc --output-format csv -- /tmp/pmc_calib ) > gpurun_out/calib_${TAG}_$c.log 2>&1
done
python tools/pmc_calib_summary.py gpurun_out/calib_${TAG}_FETCH_SIZE/pmc_counter_collection.csv gpurun_out/calib_${TAG}_WRITE_SIZE/pmc_counter_collection.csv | tee gpurun_out/calib_${TAG}.txt
