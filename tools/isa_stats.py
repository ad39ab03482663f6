"""Static/dynamic instruction mix of k_chain<6,*> from the gfx950 ISA (hipcc -save-temps).
Usage: python tools/isa_stats.py        (compiles into /tmp/isa)"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs('/tmp/isa', exist_ok=True)
subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-c', '-save-temps', '-Wno-unused-value', '-fno-slp-vectorize',
                os.path.join(ROOT, 'graspnerf_amd/csrc/gnr_kernels.hip'), '-o', '/dev/null'], cwd='/tmp/isa',
               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
txt = open('/tmp/isa/gnr_kernels-hip-amdgcn-amd-amdhsa-gfx950.s').read()
for kern in ('k_chainILi6ELb0ELb0', 'k_chainILi6ELb1ELb0'):
    m = re.search(r'\n(_ZN3gnr7%sEEEvNS_9ChainArgsE):.*?\n\s*s_endpgm' % kern, txt, re.S)
    body = m.group(0).split('\n')
    # inner loops = regions between a label and the backward branch to it
    labels = {}
    for i, l in enumerate(body):
        mm = re.match(r'^(\.LBB\d+_\d+):', l.strip())
        if mm: labels[mm.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        t = l.strip()
        mm = re.match(r'^s_c?branch\S*\s+(\.LBB\d+_\d+)', t)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            loops.append((labels[mm.group(1)], i))
    def mix(lo, hi):
        c = collections.Counter()
        for l in body[lo:hi]:
            t = l.strip()
            if not t or t.startswith(('.', ';')) or t.endswith(':'): continue
            op = t.split()[0]
            if op.startswith('v_mfma'): c['mfma'] += 1
            elif op.startswith('v_'): c['valu'] += 1; c['v:' + op] += 1
            elif op.startswith('ds_'): c['lds'] += 1
            elif op.startswith(('global_', 'buffer_', 'scratch_')): c['vmem'] += 1
            elif op == 's_nop': c['s_nop'] += 1
            elif op == 's_waitcnt': c['waitcnt'] += 1
            else: c['salu'] += 1
        return c
    print('==', kern, 'lines', len(body))
    tot = mix(0, len(body))
    print('  whole kernel static:', {k: v for k, v in tot.items() if not k.startswith('v:')})
    for lo, hi in sorted(loops):
        if hi - lo < 200: continue
        c = mix(lo, hi)
        print(f'  loop lines {lo}-{hi}:', {k: v for k, v in c.items() if not k.startswith('v:')})
        top = sorted(((v, k) for k, v in c.items() if k.startswith('v:')), reverse=True)[:14]
        print('     ', ' '.join(f'{k[2:]}={v}' for v, k in top))
