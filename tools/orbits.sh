# Whole orbits through the CLI on one MI355X (BASELINE configs 3 and 5 + the thesis benchmark), default settings and exact.
#   gpurun -- 'bash tools/orbits.sh > gpurun_out/r03/orbits.log'
B=simd-gaussian-ray-tracing_amd/bin/volumetric-ray-tracer
O=tests/golden/test-objects
run() { echo "\$ volumetric-ray-tracer $*"; $B "$@" -q 2>&1 | grep -E "TIME"; }
run -f $O/monkey.obj -w 4096 --frames 360
run -f $O/teapot.obj -w 2048 --frames 360
run -f $O/teapot.obj -w 2048
run -f $O/monkey.obj -w 4096
run -f $O/cube.obj
run -g 64 -w 2048
if [ "$1" = "exact" ]; then
run -f $O/monkey.obj -w 4096 --frames 90 -r 90 --table-step 0
run -f $O/teapot.obj -w 2048 --table-step 0
fi
