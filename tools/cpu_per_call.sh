#!/bin/bash
# usage (GPU box): tools/cpu_per_call.sh <threads> <calls>   -- closed-loop plugin clients: queries/s and the CPU time the
# whole process tree burns per call (cgroup cpu.stat; the gpurun container has a CPU quota -- cpu.max -- and throttling,
# not the GPU, bounds many-thread runs there)
T=$1; C=$2
a=$(grep usage_usec /sys/fs/cgroup/cpu.stat | cut -d' ' -f2); th0=$(grep nr_throttled /sys/fs/cgroup/cpu.stat | cut -d' ' -f2)
out=$(PC_THREADS=$T PC_CALLS=$C python3 $GRAFT_REPO_ROOT/tools/plugin_clients.py 2>&1 | grep "client threads")
b=$(grep usage_usec /sys/fs/cgroup/cpu.stat | cut -d' ' -f2); th1=$(grep nr_throttled /sys/fs/cgroup/cpu.stat | cut -d' ' -f2)
echo "$out"
echo "   whole run (build included): $(( (b - a) / 1000 )) ms of CPU, throttled periods $(( th1 - th0 )); cpu.max $(cat /sys/fs/cgroup/cpu.max)"
