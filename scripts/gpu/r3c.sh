# round 3, call c: what binds the sketch kernel?  PMC of the old kernel and of the hoisted one (same workload)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c; mkdir -p $O; cd /tmp
rocprofv3 --list-avail > $O/avail.txt 2>&1
pmc() { tag=$1; lib=$2; shift 2; SEQWIN_AMD_LIB=$lib timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $O/$tag -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/$tag.log 2>&1; echo "$tag rc=$?"; }
for v in old hoistonly; do
  lib=$R/ab/libseqwin_hip_$v.so
  pmc ${v}_a $lib SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY &&
  pmc ${v}_b $lib SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU
done
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r3c'
for d in sorted(glob.glob(O+'/*_[ab]')):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            k=row['Kernel_Name'][:40]
            if 'sketch_fast' not in k: continue
            acc[k][row['Counter_Name']]+=float(row['Counter_Value'])
    for k,v in acc.items():
        print(os.path.basename(d), k, {c: int(x) for c,x in v.items()})
PY
