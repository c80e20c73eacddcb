#!/bin/bash
# Run ON THE GPU BOX: the broker's two serving-loop knobs (gather window, retire thread) against the call rate of P Python
# workers.  Each setting starts a broker of its own (the knobs are read by the broker process at start).
cd "$(dirname "$0")/../.."
for spec in "0 0" "5 0" "0 1" "5 1" "10 1" "20 1" "5 1"; do
  set -- $spec
  echo "== gather window $1 us, retire thread $2"
  MPB_BROKER_GATHER_US=$1 MPB_BROKER_RETIRE_THREAD=$2 MOIRA_PB_BROKER_NAME=knob_$1_$2_$RANDOM timeout -k 10 240 python tools/per_read_concurrency.py 1 4 16
done
