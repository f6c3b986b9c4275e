# round 5, call A: host facts of the GPU box + full-size pin of bench workloads to the compiled reference
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5a; mkdir -p $O; cd $R
{ nproc; free -g; df -h /dev/shm /tmp; cat /sys/fs/cgroup/memory.max /sys/fs/cgroup/memory.current 2>/dev/null; lscpu | head -25; ulimit -a; } > $O/host.txt 2>&1
timeout -k 10 120 python3 scripts/pin_fullsize_ref.py --workload tiny --out $O/pin_tiny.json > $O/pin_tiny.log 2>&1 || { tail -20 $O/pin_tiny.log; exit 1; }
timeout -k 10 900 python3 scripts/pin_fullsize_ref.py --workload bacteria15k -k 21 -w 200 --out $O/pin_bacteria15k.json > $O/pin_bacteria15k.log 2>&1
echo "pin rc=$?"; tail -n 30 $O/pin_bacteria15k.log
