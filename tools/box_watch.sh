#!/bin/bash
# Every 10 s: host memory, VRAM in use, the busiest processes -> $1.  Runs beside a GPU test run so that a box that goes down leaves a trail.
out=${1:-gpurun_out/box_watch.log}
while true; do
  { date +%T; free -m | sed -n 2p; rocm-smi --showmemuse --showuse 2>/dev/null | grep -E "GPU use|VRAM%|Memory Activity" | head -4;
    ps -eo pid,rss,pcpu,etime,comm --sort=-rss | head -4; } >> "$out" 2>&1
  sleep 10
done
