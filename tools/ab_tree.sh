#!/bin/bash
# A/B of two source trees on ONE box (Python-side changes): ab/prev_tree (a git worktree of the commit to compare with, its library built / copied in) = A,
# the tree itself = B; alternating default bench steps
for r in 1 2 3; do
  echo "A $(cd ab/prev_tree && timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-config5 --no-prof 2>/dev/null | tail -1 | cut -c70-130)"
  echo "B $(timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-config5 --no-prof 2>/dev/null | tail -1 | cut -c70-130)"
done
