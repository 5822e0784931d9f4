import json, sys
d = sys.argv[1]
for l in open(f'{d}/gemmcore.log'):
    if not l.startswith('{'): continue
    r = json.loads(l)
    k = r['kernel'].replace('gemm_core ', '')
    tf = {kk[:-3]: v for kk, v in r.items() if kk.endswith('_TF')}
    print(f"{k:28s}", ' '.join(f"{a}={b:7.1f}" for a, b in tf.items()), 'bitid', r['bit_identical_to_default'])
import os
for f in [x for x in ('probe_zero.log', 'probe_rand.log') if os.path.exists(f'{d}/{x}')]:
    for l in open(f'{d}/{f}'):
        if l.startswith('{'):
            r = json.loads(l); print(r['kernel'][10:], r['ms'], r['TF'])
