"""Static instruction mix of the k_chain<6,RENDER,SAVE,USEVIS> kernels from the gfx950 ISA (hipcc -save-temps), per loop.
Usage: python tools/isa_stats.py [--no-compile] [--kern REGEX] [--flags "..."] [--top N]
Compiles graspnerf_amd/csrc/gnr_kernels.hip into /tmp/isa with build.sh's flags and prints, for each selected kernel: registers /
scratch / LDS from the metadata, the whole-body mix, and for every loop of >= 150 lines (phase 1 = first view loop, phase 2 =
second view loop; the tile loop encloses both) the mix and the most frequent VALU opcodes.  `dyn/tile` weighs the view loops by V.
The output for the product build is kept under profiles/ (rNN_isa_stats.txt)."""
import argparse, collections, os, re, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument('--no-compile', action='store_true')
ap.add_argument('--kern', default=r'k_chainILi6ELb[01]ELb0ELb0E')
ap.add_argument('--flags', default='')
ap.add_argument('--src', default='graspnerf_amd/csrc/gnr_kernels.hip')
ap.add_argument("--top", type=int, default=16); ap.add_argument("--min-lines", type=int, default=150)
ap.add_argument('--dir', default='/tmp/isa')
a = ap.parse_args()
os.makedirs(a.dir, exist_ok=True)
base = os.path.splitext(os.path.basename(a.src))[0]
asm = os.path.join(a.dir, base + '-hip-amdgcn-amd-amdhsa-gfx950.s')
if not a.no_compile:
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-c', '-save-temps', '-Wno-unused-value', '-fno-slp-vectorize']
                   + a.flags.split() + [os.path.join(ROOT, a.src), '-o', '/dev/null'], cwd=a.dir,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
txt = open(asm).read()


def classify(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_'): return 'valu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')): return 'vmem'
    if op == 's_nop': return 's_nop'
    if op == 's_waitcnt': return 'waitcnt'
    return 'salu'


def mix(lines):
    c = collections.Counter()
    for l in lines:
        t = l.strip()
        if not t or t.startswith(('.', ';')) or t.endswith(':'): continue
        op = t.split()[0]
        k = classify(op)
        c[k] += 1
        if k in ('valu', 'mfma'): c['v:' + op] += 1
        if op.startswith('scratch_'): c['scratch'] += 1
    return c


def show(c):
    return {k: v for k, v in c.items() if not k.startswith('v:')}


for m in re.finditer(r'\n(_ZN3gnr\d+(%s)\w*):.*?\n\.Lfunc_end\d+:' % a.kern, txt, re.S):
    name = m.group(1)
    body = m.group(0).split('\n')
    V = int(re.search(r'ILi(\d+)E', name).group(1)) if re.search(r'ILi(\d+)E', name) else 1
    meta = re.search(r'\.amdhsa_kernel %s\b(.*?)\.end_amdhsa_kernel' % re.escape(name), txt, re.S)
    info = {}
    if meta:
        for key in ('next_free_vgpr', 'next_free_sgpr', 'accum_offset', 'private_segment_fixed_size', 'group_segment_fixed_size'):
            mm = re.search(r'\.amdhsa_%s\s+(\S+)' % key, meta.group(1))
            if mm: info[key] = mm.group(1)
    labels = {}
    for i, l in enumerate(body):
        mm = re.match(r'^(\.LBB\d+_\d+):', l.strip())
        if mm: labels[mm.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        mm = re.match(r'^s_c?branch\S*\s+(\.LBB\d+_\d+)', l.strip())
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            loops.append((labels[mm.group(1)], i))
    loops = sorted(set(loops))
    print('==', name, '| lines', len(body), '|', info)
    tot = mix(body)
    print('  whole kernel static:', show(tot))
    big = [(lo, hi) for lo, hi in loops if hi - lo >= a.min_lines]
    # inner view loops = big loops that contain no other big loop; the tile loop is the outermost
    inner = [(lo, hi) for lo, hi in big if not any(l2 > lo and h2 < hi for l2, h2 in big)]
    outer = [(lo, hi) for lo, hi in big if (lo, hi) not in inner]
    dyn = collections.Counter()
    for lo, hi in big:
        c = mix(body[lo:hi])
        kind = 'view loop' if (lo, hi) in inner and outer else 'loop'
        print(f'  {kind} lines {lo}-{hi}:', show(c))
        top = sorted(((v, k) for k, v in c.items() if k.startswith('v:')), reverse=True)[:a.top]
        print('     ', ' '.join(f'{k[2:]}={v}' for v, k in top))
    if outer and inner:
        lo, hi = max(outer, key=lambda t: t[1] - t[0])
        c_outer = mix(body[lo:hi])
        for k, v in c_outer.items(): dyn[k] += v
        for l2, h2 in inner:
            if l2 > lo and h2 < hi:
                for k, v in mix(body[l2:h2]).items(): dyn[k] += (V - 1) * v
        print(f'  dyn/tile (view loops x{V}):', show(dyn))
        top = sorted(((v, k) for k, v in dyn.items() if k.startswith('v:')), reverse=True)[:a.top + 8]
        print('     ', ' '.join(f'{k[2:]}={v}' for v, k in top))
