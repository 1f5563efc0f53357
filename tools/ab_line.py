#!/usr/bin/env python3
"""One-line summary of a bench.py JSON line read from stdin: tools/ab_line.py <label>"""
import json, sys
line = [l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]
d = json.loads(line)
print('%-45s %.4f ms  %.3e  contacts %.0f  envs/CU %s  lds %s  status %s' % (
    sys.argv[1] if len(sys.argv) > 1 else '', d['roofline']['avg_launch_ms'], d['value'], d['contacts_per_env'],
    d['config'].get('resident_envs_per_cu'), d['config'].get('lds_bytes_per_env'), d['status_flags']))
