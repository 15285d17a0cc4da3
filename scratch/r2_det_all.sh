#!/bin/bash
# bitwise run-to-run reproducibility of every solve-kernel family (GPU box)
for n in 10 20 30 40 60 80; do REPS=10 python scratch/r2_det_trk.py $n 2>&1 | tail -1; done
for cfg in "20 3" "20 5" "30 3" "40 3" "40 5" "60 3" "60 5" "80 3" "80 5" "70 5"; do echo -n "lmpc $cfg: "; REPS=10 python scratch/r2_n80.py $cfg 2>&1 | tail -1; done
