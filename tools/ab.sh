#!/usr/bin/env bash
# A/B timing of two builds of libazsp.so on ONE box (devices differ by several per cent, so never compare across
# gpurun calls):  bash tools/ab.sh <other.so> <command...>   runs the command alternately with the in-tree build (A)
# and with AZSP_LIB=<other.so> (B), AB_ROUNDS (default three) rounds.
OTHER=$1; shift
for i in $(seq 1 ${AB_ROUNDS:-3}); do
  echo "--- A (in-tree) round $i"; "$@" 2>&1 | tail -${AB_TAIL:-3}
  echo "--- B ($OTHER) round $i"; AZSP_LIB=$OTHER "$@" 2>&1 | tail -${AB_TAIL:-3}
done
