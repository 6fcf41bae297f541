#!/usr/bin/env bash
# one evaluation-cache run of the headline workload per argument "groups:precision:gates" (gates "-" = default), on ONE box
for spec in "$@"; do IFS=: read g prec gates <<< "$spec"
  echo "== groups $g $prec gates $gates"
  if [ "$gates" != "-" ]; then export AZSP_RESNET_GATES=$gates; else unset AZSP_RESNET_GATES; fi
  python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-clock-probe --single --precision $prec --eval-cache-headline --groups $g ${EXTRA:-} 2>/dev/null |
    python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('eps/s %.0f ms/step %.0f exec %d of %d (%.2fx) frac %.3f launch %.3f ms boards/launch %.0f conc %.2f share %.2f tree_ms %.0f' % (d['value'], d['ms_per_step'], d['evals_executed'], d['evals'], d['evals']/d['evals_executed'], r['frac'], r['avg_launch_ms'], r['boards_per_launch'], r['concurrent_launches'], r['infer_share_of_wall'], d['tree']['select_ms']))"
done
