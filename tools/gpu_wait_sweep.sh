#!/bin/bash
# the host's polling waits: the kernel's timer slack (default 50 us on top of every sleep) against 1 us, alternating runs
cd /root/repo
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --cpu-contigs 0 > gpurun_out/ws_$tag.json 2>/dev/null; python - <<P
import json
try:
    j=json.load(open("gpurun_out/ws_$tag.json")); print("$tag", round(j["ms_per_step"],2), "ms", round(j["host"]["process_cpu_ms_per_step"],1), "CPU-ms")
except Exception as e: print("$tag", "failed", e)
P
}
for i in 1 2 3 4; do
run default_$i A=1
run slack1us_$i HS_TIMER_SLACK_NS=1000
run slack1us_sleep15_$i HS_TIMER_SLACK_NS=1000 HS_WAIT_SLEEP_US=15
done
