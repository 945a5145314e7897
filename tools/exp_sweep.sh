export TMPDIR=/tmp
# after "the table is sized last" (round 6): the C4 leg's slots per run and speed, round 5's library beside it; then the headline
AB_REPS=2 AB_LEGS=c4_ms_doc bash tools/ab.sh 2>&1 | cut -c1-900
