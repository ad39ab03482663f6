"""Attribute the VALU instructions of one kernel to source lines (hipcc -gline-tables-only -save-temps), per loop.
Usage: python tools/isa_attrib.py [--no-compile] [--kern MANGLED-REGEX] [--flags "..."]
Every instruction carries the innermost inlined source location (.loc), so the buckets below name the helper an instruction
belongs to (split2, elu, make_taps, ...) or the block of k_chain's body it was written in.  Weighted by V for the view loops."""
import argparse, collections, os, re, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument('--no-compile', action='store_true')
ap.add_argument('--kern', default=r'k_chainILi6ELb0ELb0ELb0E')
ap.add_argument('--flags', default='')
ap.add_argument('--dir', default='/tmp/isag')
ap.add_argument('--lines', action='store_true', help='per source line instead of per bucket')
a = ap.parse_args()
os.makedirs(a.dir, exist_ok=True)
src = os.path.join(ROOT, 'graspnerf_amd/csrc/gnr_kernels.hip')
asm = os.path.join(a.dir, 'gnr_kernels-hip-amdgcn-amd-amdhsa-gfx950.s')
if not a.no_compile:
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-c', '-save-temps', '-gline-tables-only', '-Wno-unused-value',
                    '-fno-slp-vectorize'] + a.flags.split() + [src, '-o', '/dev/null'], cwd=a.dir,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
txt = open(asm).read()
files = {int(m.group(1)): m.group(2) for m in re.finditer(r'\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', txt)}
srclines = open(src).read().split('\n')


# buckets: the function (or the marked block of k_chain) a source line of gnr_kernels.hip lies in
def build_buckets():
    b = {}
    cur = None
    fn = re.compile(r'^(?:template\s*<[^>]*>\s*)?(?:DEV|__global__|static|inline|constexpr)\b.*?\b(\w+)\s*\(')
    for i, l in enumerate(srclines, 1):
        m = fn.match(l.strip())
        if m and not l.strip().endswith(';'): cur = m.group(1)
        mm = re.search(r'// ---- (.*)$', l) or re.search(r'// =+ (.*)$', l)
        if cur == 'k_chain' and mm: b[('blk', i)] = mm.group(1)[:50]
        b[i] = cur
    return b


B = build_buckets()
blk_marks = sorted(i for k, i in [k for k in B if isinstance(k, tuple)])


def bucket(fileno, line):
    f = files.get(fileno, '?')
    if f != 'gnr_kernels.hip': return f
    fnname = B.get(line)
    if fnname == 'k_chain':
        prev = [i for i in blk_marks if i <= line]
        return 'k_chain: ' + (B[('blk', prev[-1])] if prev else 'prologue')
    return fnname or '?'


m = re.search(r'\n(_ZN3gnr\d+(%s)\w*):.*?\n\.Lfunc_end\d+:' % a.kern, txt, re.S)
body = m.group(0).split('\n')
V = int(re.search(r'ILi(\d+)E', m.group(1)).group(1))
labels, loops = {}, []
for i, l in enumerate(body):
    mm = re.match(r'^(\.LBB\d+_\d+):', l.strip())
    if mm: labels[mm.group(1)] = i
for i, l in enumerate(body):
    mm = re.match(r'^s_c?branch\S*\s+(\.LBB\d+_\d+)', l.strip())
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i: loops.append((labels[mm.group(1)], i))
big = sorted(set((lo, hi) for lo, hi in loops if sum(1 for l in body[lo:hi] if l.strip().startswith('v_')) >= 150))
inner = [(lo, hi) for lo, hi in big if not any(l2 > lo and h2 < hi for l2, h2 in big)]
outer = max(big, key=lambda t: t[1] - t[0])
cnt = collections.defaultdict(collections.Counter)
cur = (0, 0)
for i, l in enumerate(body):
    t = l.strip()
    mm = re.match(r'^\.loc\s+(\d+)\s+(\d+)', t)
    if mm: cur = (int(mm.group(1)), int(mm.group(2))); continue
    if not t or t.startswith(('.', ';')) or t.endswith(':'): continue
    op = t.split()[0]
    if not op.startswith('v_') or op.startswith('v_mfma'): kind = 'mfma' if op.startswith('v_mfma') else None
    else: kind = 'valu'
    if kind is None: continue
    region = 'outside tile loop'
    if outer[0] <= i < outer[1]: region = 'per tile'
    for n, (lo, hi) in enumerate(inner):
        if lo <= i < hi: region = f'view loop {n + 1}'
    key = (f'{files.get(cur[0], "?")}:{cur[1]} {srclines[cur[1] - 1].strip()[:70] if files.get(cur[0]) == "gnr_kernels.hip" else ""}'
           if a.lines else bucket(*cur))
    cnt[region][(kind, key)] += 1
tot = collections.Counter()
for region in sorted(cnt):
    w = V if region.startswith('view loop') else 1
    nv = sum(v for (k, _), v in cnt[region].items() if k == 'valu')
    nm = sum(v for (k, _), v in cnt[region].items() if k == 'mfma')
    print(f'== {region}: {nv} VALU, {nm} MFMA static (x{w} per tile)')
    for (k, key), v in sorted(cnt[region].items(), key=lambda t: -t[1]):
        if k == 'valu' and (v >= 3 or not a.lines): print(f'   {v:5d}  {key}')
        if k == 'valu' and region != 'outside tile loop': tot[key] += v * w
print('== per tile, all regions:', sum(tot.values()), 'VALU')
for key, v in tot.most_common(40): print(f'   {v:6d}  {key}')
