# round 4, call Z: the multi-device and gzip fuzz campaigns after the last changes (rows read by the slices' node sort; raw file buffer, 64-byte packer)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
bash scripts/gpu/fuzz.sh r4z_multi ${1:-180} multi && bash scripts/gpu/fuzz.sh r4z_gz ${1:-180} gz
