#!/bin/bash
# Shader clock / power of the GPU while the default bench runs (rocm-smi samples every 0.5 s) -> gpurun_out/clocks.txt
mkdir -p gpurun_out
(python bench.py --steps 12 --cpu-images 0 --companion-steps 0 > gpurun_out/bench_clk.json 2> gpurun_out/bench_clk.err) &
BPID=$!
: > gpurun_out/clocks.txt
while kill -0 $BPID 2>/dev/null; do
  echo "t=$(date +%s.%N | cut -c1-14) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | tr -s ' ' | tr '\n' ';')" >> gpurun_out/clocks.txt
  sleep 0.5
done
wait $BPID
echo "bench exit $?"; tail -40 gpurun_out/clocks.txt | cut -c1-240
