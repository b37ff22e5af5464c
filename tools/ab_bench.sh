#!/bin/bash
# A/B of the shipped library against variant builds on one box: tools/ab_bench.sh <variant.so>... (alternating, two rounds)
pick='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["roofline"]["kernel_ms_avg"])'
for i in 1 2; do
  python bench.py --steps 20 --warmup 3 --no-cpu $AB_ARGS 2>/dev/null | python -c "$pick" main
  for v in "$@"; do
    GOBLIN_HIP_LIB=$v python bench.py --steps 20 --warmup 3 --no-cpu $AB_ARGS 2>/dev/null | python -c "$pick" "$(basename $v)"
  done
done
