"""Per-kernel summary of a rocprofv3 --kernel-trace run (rocpd sqlite output): python scripts/prof_summary.py gpurun_out/prof_x [steps]"""
import glob, re, sqlite3, sys
db = glob.glob(sys.argv[1] + '/*/*_results.db')[0]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
sy = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
def short(n):
    m = re.search(r'(k_[a-z0-9_]+|sketch_[a-z]+_kernel(ILi\d+E)?)', n)
    if m: return m.group(1)
    for key in ('onesweep_histograms', 'onesweep', 'radix_sort', 'scan', 'reduce', 'partition', 'select'):
        if key in n: return 'rocprim::' + key + ('<u64>' if 'ImLb' in n or 'Im' in n[-60:] else '')
    return n[:50]
q = f"select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start) from {kd} d join {sy} s on d.kernel_id=s.id group by s.kernel_name order by 4 desc"
tot = 0
print(f"{'kernel':48s} {'calls/step':>10s} {'avg_us':>9s} {'ms/step':>8s}")
for n, cnt, avg, t in c.execute(q):
    tot += t
    if t / 1e6 / steps < 0.005: continue
    print(f"{short(n):48s} {cnt/steps:10.1f} {avg/1e3:9.1f} {t/1e6/steps:8.3f}")
print("total kernel ms/step", tot / 1e6 / steps)
