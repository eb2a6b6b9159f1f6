#!/bin/bash
# Broader bitwise-parity sweep (HIP path vs CPU oracle) than the test-suite: seeds, flags, team sizes, action kinds.
set -e
run() { timeout -k 5 280 python tools/parity_run.py "$@" --stop 2>&1 | tail -1; }
run --worlds 64 --steps 250 --act bench --seed 11
run --worlds 48 --steps 130 --act full --seed 21 --hiders 3 --seekers 3
run --worlds 48 --steps 130 --act full --seed 22 --hiders 3 --seekers 3 --flags 13
run --worlds 48 --steps 130 --act full --seed 23 --hiders 2 --seekers 3 --flags 8
run --worlds 48 --steps 130 --act full --seed 24 --hiders 1 --seekers 1 --flags 4
run --worlds 48 --steps 130 --act full --seed 25 --hiders 3 --seekers 1 --flags 2
run --worlds 40 --steps 260 --act full --seed 5 --hiders 3 --seekers 3 --flags 13
run --worlds 33 --steps 100 --act full --seed 31 --hiders 2 --seekers 2 --flags 9
# a dozen more seeds with team sizes 1..3 x 1..3 and flags 0 / 4 / 8 / 12 (grab / lock actions, every third step compared)
for s in 101 102 103 104 105 106 107 108 109 110 111 112; do
    h=$((s % 3 + 1)); k=$(((s / 3) % 3 + 1)); f=$(( ((s % 4) * 4 + (s % 2) * 8) % 16 ))
    run --worlds 96 --steps 250 --act full --seed $s --hiders $h --seekers $k --flags $f --every 3
done
