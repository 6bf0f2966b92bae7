#!/bin/bash
# defaults vs fused FFT dim0 pass at every volume size (one view, 15^3 PSF); config 1; smoke; GPU suite
set -e
mkdir -p gpurun_out/r03s
export AB_NO_FFT=1
for s in "32 32 32" "64 64 64" "96 96 96" "128 128 128" "160 160 160" "192 192 192" "256 256 256" "384 384 384" "512 512 512" "64 512 512" "16 1920 1920"; do
  echo "== $s" >> gpurun_out/r03s/shapes.txt
  AB_SHAPE="$s" python3 tools/sweep.py "" "MVN_DIM0_DIRECT=0" "MVN_DIM0_DIRECT=0,MVN_NYQ_PACKED=0" >> gpurun_out/r03s/shapes.txt 2>&1
done
echo shapes done
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03s/smoke.log 2>&1
python3 bench.py --config 1 > gpurun_out/r03s/config1.json 2> gpurun_out/r03s/config1.err
MVN_DIM0_DIRECT=0 python3 bench.py --config 1 > gpurun_out/r03s/config1_fft.json 2> gpurun_out/r03s/config1_fft.err
python3 bench.py > gpurun_out/r03s/bench.json 2> gpurun_out/r03s/bench.err
echo bench done
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03s/gputest.log 2>&1
tail -3 gpurun_out/r03s/gputest.log
