# round 5, call T: why the mapped ingest is slow inside bench.py (15 Gbp/s) and fast in the A/B script (30): torch in the process? the reference's runs before it?
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5t; mkdir -p $O; cd $R
echo plain; timeout -k 10 200 python3 tests/tools/e2e_ingest_ab.py 2048 32 2>&1 | grep -v amdgpu
echo torch; AB_TORCH=1 timeout -k 10 200 python3 tests/tools/e2e_ingest_ab.py 2048 32 2>&1 | grep -v amdgpu
echo ref; AB_REF=1 timeout -k 10 300 python3 tests/tools/e2e_ingest_ab.py 2048 32 2>&1 | grep -v amdgpu
