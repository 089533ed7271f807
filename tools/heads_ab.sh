#!/bin/bash
# Separate-process A/B of the style heads' stream placement (ST_HEAD_XCC*; csrc/st_cumask.hip) and of related switches:
# every variant is a fresh `bench.py --no-extra --no-cpu-baseline` process, variants interleaved, REPS rounds.
#   tools/heads_ab.sh "512 256" 2 "base:" "half:ST_HEAD_XCC4=15 ST_HEAD_XCC3=240" ...
# Prints one line per (size, variant): the it/s of every round.
SIZES="$1"; REPS="$2"; shift 2
cd "$(dirname "$0")/.."
for size in $SIZES; do
  declare -A res
  for rep in $(seq 1 "$REPS"); do
    for spec in "$@"; do
      name="${spec%%:*}"; envs="${spec#*:}"
      v=$(env $envs timeout 120 python bench.py --no-extra --no-cpu-baseline --size "$size" --steps ${STEPS:-60} --warmup 10 2>/dev/null |
          python -c "import sys,json; print('%.1f' % json.loads(sys.stdin.readline())['value'])" 2>/dev/null)
      res[$name]="${res[$name]} ${v:-fail}"
    done
  done
  for spec in "$@"; do name="${spec%%:*}"; echo "[heads_ab] size $size $name:${res[$name]}   (${spec#*:})"; done
  unset res
done
