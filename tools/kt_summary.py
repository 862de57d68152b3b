"""Per-kernel totals of a rocprofv3 --kernel-trace --stats csv, grouped by a
few name patterns.  python tools/kt_summary.py <kernel_stats.csv> [pattern ...]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pats = sys.argv[2:] or ['bn_', 'conv_igemm_kernel']
for p in pats:
  sel = [r for r in rows if p in r['Name']]
  print('%-24s calls %6d total %9.2f ms' % (p, sum(int(r['Calls']) for r in sel),
                                            sum(float(r['TotalDurationNs']) for r in sel) / 1e6))
  for r in sel:
    print('   %-58s calls %5s avg %8.1f us' % (r['Name'][26:84], r['Calls'], float(r['AverageNs']) / 1e3))
