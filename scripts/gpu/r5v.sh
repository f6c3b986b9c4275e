# round 5, call V: longer soak of the multi-device and gzip fuzz sets on the final library (400 s each)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
bash scripts/gpu/fuzz.sh r5v/fuzz_multi 400 multi && bash scripts/gpu/fuzz.sh r5v/fuzz_gz 400 gz
