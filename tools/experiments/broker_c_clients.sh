#!/bin/bash
# Run ON THE GPU BOX: the broker's capacity with C client threads (no Python in the callers), then with Python workers.
cd "$(dirname "$0")/../.."
g++ -O2 -std=c++17 -pthread -Iinclude tools/experiments/broker_c_clients.cpp moira_amd/libmoira_pb.so -Wl,-rpath,$PWD/moira_amd -o /tmp/bcc || exit 1
(PYTHONPATH=$PWD timeout -k 5 200 python -m moira_amd.broker --name cc --idle-exit 20 > /tmp/bcc_broker.log 2>&1 &)
sleep 1
timeout -k 5 150 /tmp/bcc cc 1 2 4 8 16 32
python -c "
import sys; sys.path.insert(0, '.')
from moira_amd import broker; broker.shutdown('cc')"
tail -2 /tmp/bcc_broker.log
timeout -k 10 200 python tools/per_read_concurrency.py 1 8 16
