# usage: prof_stats.sh <tag> [bench args...]   -> gpurun_out/<tag>/stats (kernel trace + stats, csv)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; T=$1; shift; O=$R/gpurun_out/$T; mkdir -p $O; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/stats.log 2>&1
echo rc=$?
