#!/usr/bin/env python3
"""Scratch (spill) traffic inside the iteration loop of one kernel shape: compiles tools/kres/kres.hip to ISA and counts, per
basic block of the loop that holds the stage-1/stage-2 FMA blocks, the scratch loads/stores.  usage: hotloop.py "2,16,8,7,7,4,2" """
import re, subprocess, sys
shape = sys.argv[1]
extra = sys.argv[2:]
src = __file__.rsplit('/', 1)[0] + '/kres.hip'
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-S', '--cuda-device-only',
                       '-DKRES_SHAPE=' + shape] + extra + [src, '-o', '/tmp/kres_hot.s'], stderr=subprocess.DEVNULL)
lines = open('/tmp/kres_hot.s').read().split('\n')
cur = None; seq = []
for l in lines:
    m = re.match(r'^(\.LBB\d+_\d+):\s*;?(.*)', l)
    if m:
        cur = dict(name=m.group(1), cm=m.group(2).strip(), fma=0, sl=0, ss=0, ds=0, n=0, bar=0, valu=0)
        seq.append(cur); continue
    if cur is None: continue
    t = l.strip()
    if t and not t.startswith(';') and not t.startswith('.'):
        cur['n'] += 1
        if t.startswith('v_fma'): cur['fma'] += 1
        if t.startswith('v_'): cur['valu'] += 1
        if t.startswith('scratch_load'): cur['sl'] += 1
        if t.startswith('scratch_store'): cur['ss'] += 1
        if t.startswith('ds_'): cur['ds'] += 1
        if t.startswith('s_barrier'): cur['bar'] += 1
big = [b for b in seq if b['fma'] >= 40 and 'Depth=2' in (b['cm'] + ' Depth=2' if b['cm'].startswith('Parent') else b['cm'])]
hdrs = {}
for b in seq:
    m = re.search(r'Header=(BB\d+_\d+) Depth=2', b['cm'])
    if m: hdrs.setdefault(m.group(1), []).append(b)
for h, bs in hdrs.items():
    head = [b for b in seq if b['name'] == '.L' + h]
    allb = head + bs
    f = sum(b['fma'] for b in allb)
    if f < 100: continue
    print('loop', h, 'blocks', len(allb), 'instr', sum(b['n'] for b in allb), 'fma', f, 'valu', sum(b['valu'] for b in allb), 'ds', sum(b['ds'] for b in allb),
          'barriers', sum(b['bar'] for b in allb), 'scratch loads', sum(b['sl'] for b in allb), 'stores', sum(b['ss'] for b in allb))
    hot = [b for b in allb if b['fma'] >= 40]
    print('  FMA blocks:', [(b['name'], b['fma'], 'sl', b['sl'], 'ss', b['ss']) for b in hot])
tot_l = sum(b['sl'] for b in seq); tot_s = sum(b['ss'] for b in seq)
print('kernel total scratch loads', tot_l, 'stores', tot_s)
