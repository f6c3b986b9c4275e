# round 5, call AL: soak of the default and the multi-device fuzz sets on the round's last library (280 s each)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
bash scripts/gpu/fuzz.sh r5al/fuzz 280 && bash scripts/gpu/fuzz.sh r5al/fuzz_multi 280 multi
