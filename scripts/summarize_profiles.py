"""Turn rocprofv3 CSV output under gpurun_out/ into the small text summaries committed under profiles/."""
import collections, csv, glob, json, re, sys
tag = sys.argv[1]          # e.g. r1f
out_prefix = sys.argv[2]   # e.g. profiles/r01f_v4
def short(n):
    m = re.search(r'(k_\w+|sketch_\w+(<\d+(, ?\d+)?>)?)', n)
    if m: return m.group(1)
    if 'radix_sort' in n: return 'rocprim::radix_sort_onesweep'
    if 'scan' in n: return 'rocprim::scan'
    return n[:40]
ks = glob.glob(f'gpurun_out/prof_{tag}/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(ks)))
with open(out_prefix + '_kernel_stats.txt', 'w') as f:
    f.write("rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 1   (salmonella500; 6 passes incl. warm-up)\n")
    f.write(f"{'kernel':34s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>9s} {'pct':>6s}\n")
    for r in rows[:24]:
        f.write(f"{short(r['Name']):34s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.1f} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['Percentage']):6.2f}\n")
def agg(path, names):
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] in names: a[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    return a
pm = glob.glob(f'gpurun_out/pmc_{tag}/*/*counter_collection.csv')[0]
names = ['SQ_WAVES','SQ_BUSY_CYCLES','SQ_WAVE_CYCLES','SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_LDS','SQ_WAIT_ANY','SQ_WAIT_INST_ANY']
a = agg(pm, names)
bp = 2.4576e9
with open(out_prefix + '_pmc_sketch.txt', 'w') as f:
    f.write("rocprofv3 --pmc SQ_* -- python3 bench.py --steps 2 --warmup 1 (salmonella500, 2.4576 Gbp per launch); per-dispatch averages\n")
    for k, v in a.items():
        if 'sketch' not in k: continue
        f.write(f"{k}\n")
        for c in names:
            if c in v: f.write(f"   {c:18s} {sum(v[c])/len(v[c]):16.0f}\n")
        if 'fast' in k:
            iv = sum(v['SQ_INSTS_VALU'])/len(v['SQ_INSTS_VALU'])
            f.write(f"   => {iv/bp:.3f} VALU wave-instructions per bp = {iv/bp*64:.1f} lane-operations per bp; "
                    f"WAIT_ANY/WAVE_CYCLES = {sum(v['SQ_WAIT_ANY'])/sum(v['SQ_WAVE_CYCLES']):.2f}\n")
fe = agg(glob.glob(f'gpurun_out/pmc_{tag}_fetch/*/*counter_collection.csv')[0], ['FETCH_SIZE'])
wr = agg(glob.glob(f'gpurun_out/pmc_{tag}_write/*/*counter_collection.csv')[0], ['WRITE_SIZE'])
traffic = {}
with open(out_prefix + '_hbm_traffic.txt', 'w') as f:
    f.write("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 (salmonella500)\n")
    f.write("FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of the bytes of wide\ncoalesced reads, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact.  Per-dispatch averages:\n")
    for k in fe:
        fk = sum(fe[k]['FETCH_SIZE'])/len(fe[k]['FETCH_SIZE']); wk = sum(wr.get(k,{}).get('WRITE_SIZE',[0]))/max(1,len(wr.get(k,{}).get('WRITE_SIZE',[0])))
        f.write(f"  {k:34s} n={len(fe[k]['FETCH_SIZE']):3d} FETCH={fk:11.0f} KiB WRITE={wk:11.0f} KiB -> read {2*fk*1024/1e6:8.1f} MB (corrected) write {wk*1024/1e6:8.1f} MB\n")
        if 'sketch_fast' in k:
            traffic = dict(kernel=k, workload='salmonella500', k=21, w=200, fetch_kib=fk, write_kib=wk,
                           read_bytes_corrected=2*fk*1024, write_bytes=wk*1024, hbm_bytes_per_launch=2*fk*1024+wk*1024,
                           note='FETCH_SIZE and WRITE_SIZE from separate rocprofv3 --pmc passes; FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md',
                           source=out_prefix + '_hbm_traffic.txt')
fast = [v for k, v in a.items() if 'sketch_fast' in k]
if fast and traffic:
    traffic['valu_wave_insts_per_launch'] = sum(fast[0]['SQ_INSTS_VALU']) / len(fast[0]['SQ_INSTS_VALU'])
json.dump(traffic, open('profiles/traffic.json', 'w'), indent=1)
print(open(out_prefix + '_kernel_stats.txt').read()); print(open(out_prefix + '_pmc_sketch.txt').read()); print(traffic)
