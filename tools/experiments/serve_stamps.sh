#!/bin/bash
# Run ON THE GPU BOX: the resident server's turnaround (door word -> results, broker side) and a request's time inside its
# wave with the shader clock it ran at (k_serve stamps; experiment build), for P workers.
FL="-O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math -pthread -Iinclude -x hip"
mkdir -p /tmp/var
/opt/rocm/bin/hipcc $FL -DMPB_SERVE_STAMPS $EXTRA moira_amd/csrc/mpb_kernels.hip moira_amd/csrc/mpb_api.cpp moira_amd/csrc/mpb_broker.cpp -o /tmp/var/serve.so 2>/tmp/var/serve.err || { tail -5 /tmp/var/serve.err; exit 1; }
export XDG_RUNTIME_DIR=/tmp/xdg_$$; mkdir -p $XDG_RUNTIME_DIR; chmod 700 $XDG_RUNTIME_DIR
for p in ${WORKERS:-1 16}; do
  MOIRA_PB_LIB=/tmp/var/serve.so MPB_BROKER_TRACE=1 timeout -k 10 200 python3 tools/per_read_concurrency.py $p 2>&1 | tail -2
  sleep 1
  cat $XDG_RUNTIME_DIR/*/broker_*.log $XDG_RUNTIME_DIR/broker_*.log 2>/dev/null | grep -a "\[broker\]\|\[k_serve\|\[small_one" | tail -4
done
