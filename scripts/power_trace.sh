#!/bin/bash
# Power, clock and temperature of the GPU once a second while the training-step bench runs (the matrix kernels
# are bounded by the chip's power budget, not by their schedules).  bash scripts/power_trace.sh > gpurun_out/power_trace.txt
timeout -k 10 150 python3 bench.py --steps ${STEPS:-250} --warmup 5 --no-cpu-baseline --no-kernel-timing > /tmp/power_bench.json 2>/dev/null &
pid=$!
for i in $(seq 1 ${SAMPLES:-28}); do
  sleep 1
  echo "t=${i}s $(rocm-smi -d 0 --showpower --showclocks --showtemp 2>/dev/null | grep -E 'Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge|hotspot)' | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';')"
  kill -0 $pid 2>/dev/null || break
done
wait $pid
echo "bench: $(python3 -c 'import json; d=json.load(open("/tmp/power_bench.json")); print(d["ms_per_step"], "ms/step")')"
