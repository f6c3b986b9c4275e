# round 5, call R: host ingest reading its files by read() / mmap / mmap + MAP_POPULATE from many workers (16-CPU quota)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5r; mkdir -p $O; cd $R
for n in 64 32 16; do timeout -k 10 300 python3 tests/tools/e2e_ingest_ab.py 2048 $n > $O/ab$n.log 2>&1; grep -v amdgpu $O/ab$n.log; done
