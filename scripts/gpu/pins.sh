# pins.sh <tag> <spec>...: full-size pins to the compiled reference (scripts/pin_fullsize_ref.py), one after the other; a spec is
# workload:k:w[:size-from].  Outputs gpurun_out/<tag>/pin_<workload>_k<k>_w<w>.{json,log}; merge locally with scripts/merge_pins.py.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R; shift
for spec in "$@"; do
  IFS=: read wl k w from <<< "$spec"
  n=pin_${wl}_k${k}_w${w}
  timeout -k 10 560 python3 scripts/pin_fullsize_ref.py --workload $wl -k $k -w $w ${from:+--size-from "$from"} --out $O/$n.json > $O/$n.log 2>&1
  rc=$?; echo "$n rc=$rc"; tail -n 4 $O/$n.log
  [ $rc -eq 0 ] || exit $rc
done
