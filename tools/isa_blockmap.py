"""Where a kernel's scratch traffic sits: per basic block of the gfx950 ISA (a -save-temps .s file of tools/isa_one.sh or
tools/scratch_report.py) the instruction count, scratch stores / loads and MFMAs, with the loop annotations of the assembler.
A spill in a prologue or once per tile is noise; one inside a view loop is a cost.   python tools/isa_blockmap.py <mangled name> [file.s]"""
import re, sys
name = sys.argv[1]
path = sys.argv[2] if len(sys.argv) > 2 else '/tmp/isa_one/one-hip-amdgcn-amd-amdhsa-gfx950.s'
t = open(path).read()
i = t.index(name + ':')
body = t[i:t.index('.Lfunc_end', i)].split('\n')
cur, stats, order = 'entry', {'entry': [0, 0, 0, 0, '']}, ['entry']
for l in body:
    m = re.match(r'^(\.LBB\d+_\d+):(.*)', l)
    if m:
        cur = m.group(1); order.append(cur); stats[cur] = [0, 0, 0, 0, m.group(2).strip()]
        continue
    s = l.strip()
    if not s or s.startswith(';') or s.startswith('.'):
        continue
    st = stats[cur]
    st[0] += 1
    st[1] += s.startswith('scratch_store'); st[2] += s.startswith('scratch_load'); st[3] += s.startswith('v_mfma')
print(f'{name}: {sum(v[0] for v in stats.values())} instructions, {sum(v[1] for v in stats.values())} scratch stores, {sum(v[2] for v in stats.values())} scratch loads')
for b in order:
    st = stats[b]
    if st[1] + st[2] or st[3] >= 8:
        print(f'  {b:12s} insts {st[0]:5d}  scratch st {st[1]:3d} ld {st[2]:3d}  mfma {st[3]:3d}  {st[4][:80]}')
