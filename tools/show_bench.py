#!/usr/bin/env python3
"""Print the per-kernel and host-phase figures of bench.py JSON lines (files given on the command line)."""
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d['value']), 'fps', round(d['ms_per_step'], 2), 'ms/step')
    print('  kernels avg_us:', {k.replace('k_ekf_', 'e_').replace('k_', ''): round(v['avg_us'], 1) for k, v in d['kernels'].items() if v['launches']})
    print('  host ms/step  :', d['host_phases_ms_per_step'])
