"""prints the records of a bench.py JSON line as a table (dev tool): python tools/show_bench.py gpurun_out/b.log"""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
def show(r, name):
    rf = r['roofline']
    print("%-26s value=%.3e ms/step=%8.2f %-14s %8.2fms frac=%.3f line=%s nodes/s=%.2e build=%ss idx=%.1fGB hits=%s" % (
        name, r['value'], r['ms_per_step'], rf['kernel'], rf['kernel_ms'], rf['frac'], ('%.3f' % rf['line_granular']['frac']) if 'line_granular' in rf else '  -  ',
        rf['units_per_launch'] / rf['kernel_ms'] * 1e3, r['config']['index_build_s'], r['config']['index_device_bytes'] / 1e9, r.get('hits')))
    if 'symbols_until_one_row' in r: print("    depth:", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in r['symbols_until_one_row'].items() if k != 'what'})
    if 'cpu_baseline' in r:
        c = r['cpu_baseline']; print("    cpu: %.3e q/s on %d cores, 1 thread %.3e, eff %.2f, match=%s" % (c['value'], c['cores'], c['single_thread']['value'], c['parallel_efficiency'], c['gpu_results_match_on_sample']))
    if 'exchange' in r: print("    exchange:", r['exchange'])
show(d, 'HEAD ' + d['config'].get('index_kind', ''))
if 'secondary' in d: show(d['secondary'], 'SECONDARY')
for r in d.get('records', []): show(r, r['id'])
