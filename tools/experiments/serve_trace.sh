#!/bin/bash
# Run ON THE GPU BOX: the broker process itself under rocprofv3 --kernel-trace --stats while 8 workers call through it: the
# resident kernel's launches (one per 100 ms lifetime while calls come) and their durations.
export TMPDIR=/tmp
cd /tmp
N=prof_$$
D=$GRAFT_REPO_ROOT/gpurun_out/serve_trace
rm -rf $D; mkdir -p $D
PYTHONPATH=$GRAFT_REPO_ROOT rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 -m moira_amd.broker --device 0 --name $N --slots 64 --idle-exit 2 > $D/broker.log 2>&1 &
BP=$!
sleep 8
cd $GRAFT_REPO_ROOT
MOIRA_PB_BROKER_NAME=$N timeout -k 10 120 python3 tools/per_read_concurrency.py 8 2>&1 | tail -1
wait $BP
cat $D/*/*kernel_stats.csv | cut -c1-160
