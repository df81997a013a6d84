#!/bin/bash
# alternating A/B of two builds / settings on a GPU box: tools/ab.sh "ENV=.. ENV=.." "ENV=.." NAME... (tools/dev.py rate, 16 launches each, three rounds)
A=$1; B=$2; shift 2
for i in 1 2 3; do
  env $A timeout -k 10 100 python3 tools/dev.py rate "$@" --launches 16 2>&1 | grep Mrays | cut -c1-48 | sed 's/^/A /'
  env $B timeout -k 10 100 python3 tools/dev.py rate "$@" --launches 16 2>&1 | grep Mrays | cut -c1-48 | sed 's/^/B /'
done
