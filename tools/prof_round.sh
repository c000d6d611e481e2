# Every profile of a round (run on the GPU box): bash tools/prof_round.sh r03 — the name the round-2 VERDICT used; the work is in prof_all.sh
exec bash "$(dirname "$0")/prof_all.sh" "$@"
