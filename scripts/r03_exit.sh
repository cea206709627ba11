#!/bin/bash
t() { for i in 1 2 3; do s=$(date +%s%N); "$@" 2> /tmp/ep.err; e=$(date +%s%N); echo "wall $(( (e - s) / 1000000 )) ms leaving->gone $(( e / 1000000 - $(grep -o '[0-9]*$' /tmp/ep.err) )) ms : $*"; done; }
P=./scripts/exit_probe.bin
t $P 64 0 0 0 1
t $P 64 0 1 0 1 0
t $P 64 0 4 0 1 0
t $P 64 0 16 0 1 0
t $P 64 0 16 0 1 1
t $P 64 0 16 0 1 2
t $P 64 0 16 0 1 3
t $P 64 400 16 0 1 0
t $P 64 400 16 0 1 2
t $P 64 160 16 0 1 0
