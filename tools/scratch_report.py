"""Registers / scratch of every kernel of the library from a -save-temps compile (see tools/isa_stats.py for the per-loop mix):
python tools/scratch_report.py [--no-compile] [--all]   -> kernels with scratch (or all), largest first."""
import argparse, os, re, subprocess, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument('--no-compile', action='store_true'); ap.add_argument('--all', action='store_true'); ap.add_argument('--dir', default='/tmp/isa')
a = ap.parse_args()
os.makedirs(a.dir, exist_ok=True)
if not a.no_compile:
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-c', '-save-temps', '-Wno-unused-value', '-fno-slp-vectorize',
                    os.path.join(ROOT, 'graspnerf_amd/csrc/gnr_kernels.hip'), '-o', '/dev/null'], cwd=a.dir, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
t = open(os.path.join(a.dir, 'gnr_kernels-hip-amdgcn-amd-amdhsa-gfx950.s')).read()
filt = shutil.which('c++filt') or shutil.which('llvm-cxxfilt')
rows = []
for m in re.finditer(r'- \.agpr_count:.*?\.wavefront_size', t, re.S):
    blk = m.group(0)
    g = lambda k: re.search(r'\.' + k + r':\s+(\S+)', blk).group(1)
    rows.append((int(g('private_segment_fixed_size')), g('name'), int(g('vgpr_count')), int(g('agpr_count')), int(g('vgpr_spill_count'))))
print('scratch_B  vgpr  agpr  spills  kernel')
for sc, name, v, ag, sp in sorted(rows, reverse=True):
    if sc or a.all:
        d = subprocess.run([filt, name], capture_output=True, text=True).stdout.strip() if filt else name
        print(f'{sc:9d} {v:5d} {ag:5d} {sp:7d}  {d[:130]}')
print(f'{sum(1 for r in rows if r[0])} of {len(rows)} kernels use scratch')
