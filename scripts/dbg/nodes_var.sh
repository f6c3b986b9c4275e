cd $GRAFT_REPO_ROOT
for v in default 1024_0_4 1024_0_6 1024_0_8; do
  if [ $v = default ]; then unset SEQWIN_AMD_LIB; else export SEQWIN_AMD_LIB=$GRAFT_REPO_ROOT/seqwin_amd/csrc/build/var/libvar_$v.so; fi
  python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err && python3 -c "
import json; d=json.load(open('/tmp/b.json')); print('$v', d['value'], d['ms_per_step'], d['stages_ms']['nodes_ms'], d['parity'])"
done
