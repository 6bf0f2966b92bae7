mkdir -p gpurun_out/r03x
export AB_NO_FFT=1
for s in "64 64 64" "128 128 128" "192 192 192" "256 256 256"; do
  echo "== $s" >> gpurun_out/r03x/graph.txt
  AB_SHAPE="$s" python3 tools/sweep.py "" "MVN_GRAPH=1" "MVN_GRAPH=0" "MVN_GRAPH=1" >> gpurun_out/r03x/graph.txt 2>&1
done
grep -E "^==|view-iter" gpurun_out/r03x/graph.txt | cut -c1-60
FUZZ_FIXED=1 FUZZ_DEEP=1 FUZZ_MAX_VOXELS=60000000 timeout -k 10 500 python3 tools/fuzz_shapes.py 40 21 > gpurun_out/r03x/fuzz_fixed_final.txt 2>&1
tail -1 gpurun_out/r03x/fuzz_fixed_final.txt
FUZZ_DEEP=1 timeout -k 10 300 python3 tools/fuzz_shapes.py 60 22 > gpurun_out/r03x/fuzz_final.txt 2>&1
tail -1 gpurun_out/r03x/fuzz_final.txt
