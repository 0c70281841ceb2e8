"""Print a compact table of a bench.py JSON line (file argument or stdin)."""
import json
import sys

src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin
d = None
for line in src:
    if line.startswith('{"metric"'):
        d = json.loads(line)
if d is None:
    sys.exit("no bench line found")
print('value %.3e rows/s  ms/step %.4f  hits %s  cpu_hits %s' % (d['value'], d['ms_per_step'], d['config']['hits'],
                                                                 d['config'].get('cpu_hits')))
r = d['roofline']
print('primary %-18s ms %.4f frac %.3f | cold ms %.4f frac %.3f | kernel MB %.1f alg MB %.1f eff GB/s %.0f' % (
    r['kernel'], r['kernel_ms'], r['frac'], r.get('kernel_ms_l3_cold', 0), r.get('frac_l3_cold', 0),
    r['kernel_bytes_per_launch'] / 1e6, r['algorithmic_bytes_per_launch'] / 1e6, r['effective_gbs']))
if r.get('frac_by_traffic'):
    print('   by measured HBM traffic: %.1f MB per launch, frac %.3f' % (r['traffic'] / 1e6, r['frac_by_traffic']))
if 'cpu_baseline' in d:
    print('cpu 1 core %.3e rows/s; all cores (%d) %.3e' % (d['cpu_baseline']['value'], d['cpu_baseline_all_cores']['cores'],
                                                         d['cpu_baseline_all_cores']['value']))
for k, v in d.get('secondary', {}).items():
    if 'error' in v:
        print(k, v)
    elif 'kernel_ms' in v:
        print('%-26s ms %.4f frac %.3f | cold ms %.4f frac %.3f | hits %s' % (k, v['kernel_ms'], v['frac'],
              v['kernel_ms_l3_cold'], v['frac_l3_cold'], v.get('hits')))
        for kk, vv in v.get('get_with_selection', {}).items():
            print('   gather %-7s ms %.4f frac %.3f eff GB/s %.0f' % (kk, vv['ms'], vv['frac'], vv['effective_gbs']))
    elif 'queries' in v:
        print('%s: %d rows, %d columns, all queries %.3f ms' % (k, v['rows'], v['columns'], v['total_ms_all_queries']))
        print('   ' + '  '.join('%s %.3fms/%d' % (q, r['ms'], r['rows_out']) for q, r in v['queries'].items()))
    else:
        print(k, json.dumps(v)[:700])
