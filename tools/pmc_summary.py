"""Aggregate rocprofv3 --pmc CSV outputs (one run per counter set) into the per-kernel JSON kept under profiles/.
Usage: python tools/pmc_summary.py <dir with *_counter_collection.csv> <out.json>      (bench.py imports aggregate / document for its live passes)"""
import csv, glob, json, os, re, sys
from collections import defaultdict

# the counter sets, one rocprofv3 run each (tools/collect_profiles.sh; bench.py's live passes)
FWD_SETS = ["FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum TCC_MISS_sum", "SQ_INSTS_VALU SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES",
            "SQ_WAVE_CYCLES GRBM_GUI_ACTIVE", "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY", "SQ_INSTS_LDS SQ_INSTS_VMEM_RD", "SQ_WAIT_INST_ANY SQ_WAIT_ANY"]
BWD_SETS = ["FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum TCC_MISS_sum", "SQ_INSTS_VALU SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"]


def aggregate(src):
    """Per-kernel, per-counter averages over the dispatches found under `src` (directories named bwd_* hold the backward passes)."""
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(src, '**', '*counter_collection.csv'), recursive=True):
        per_dispatch = defaultdict(float)
        meta = {}
        for r in csv.DictReader(open(f)):
            key = (f, r['Dispatch_Id'], r['Counter_Name'])
            per_dispatch[key] += float(r['Counter_Value'])
            meta[(f, r['Dispatch_Id'])] = r['Kernel_Name']
        for (ff, did, cname), v in per_dispatch.items():
            name = meta[(ff, did)]
            m = re.search(r'gnr::(k_\w+(<[^>]*>)?)', name)
            if m:
                k = m.group(1)
                # the backward passes (pmc/bwd_*: tools/time_volume_bwd.py, 8 scenes) also launch inference kernels at another batch
                # size: keep only the training kernels from them
                if os.sep + 'bwd_' in ff and not ('_bwd' in k or k.startswith('k_scatter_') or re.match(r'k_chain<\d+, (true|false), true,', k)):      # k_chain<V, RENDER, SAVE, ..>
                    continue
                acc[k][cname].append(v)
    res = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in sorted(acc.items())}
    for k, cs in res.items():
        if 'FETCH_SIZE' in cs and 'WRITE_SIZE' in cs:
            cs['hbm_bytes_corrected'] = (2 * cs['FETCH_SIZE'] + cs['WRITE_SIZE']) * 1024
    return res


import hashlib
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'graspnerf_amd', 'csrc')
BWD_SOURCES = ('gnr_kernels.hip', 'gnr_bwd.inc', 'gnr_bwd_scatter.inc', 'gnr_bwd_view1_pw.inc', 'gnr_bwd_view2_pw.inc', 'gnr_bwd_geo_dual_mm.inc')


def sha16(*names):
    h = hashlib.sha256()
    for n in names:
        h.update(open(os.path.join(CSRC, n), 'rb').read())
    return h.hexdigest()[:16]


def commit_count():
    import subprocess
    try:
        return int(subprocess.check_output(['git', '-C', os.path.dirname(CSRC), 'rev-list', '--count', 'HEAD'], stderr=subprocess.DEVNULL).decode())
    except Exception:
        return None          # no history on the GPU box: tools/stamp_pmc.py sets it once the file is back in the repo


def document(res):
    """The JSON kept under profiles/ (and built in memory by bench.py's live passes)."""
    return {
        'git_commit_count': commit_count(),
        # bench.py reports recorded counters only while the stamps equal the sha256 of the kernel sources it runs (bench.py newest_pmc)
        'kernel_source_sha16': sha16('gnr_kernels.hip'), 'bwd_source_sha16': sha16(*BWD_SOURCES),
        'command': 'rocprofv3 --pmc <set> --output-format csv -- python tools/run_hot.py --iters 1   (forward kernels: B=32 scenes, 6 views, '
                   '40^3 + 512 rays) and -- python tools/time_volume_bwd.py --scenes 8 (k_*_bwd kernels of sample_volume: 8 scenes); '
                   'one run per counter set: ' + ' | '.join(FWD_SETS) + '; tools/collect_profiles.sh, bench.py live_pmc',
        'units': 'FETCH_SIZE / WRITE_SIZE in KiB as reported by rocprofv3; per-launch averages',
        'note': 'gfx950: FETCH_SIZE under-counts wide (16 B/lane) reads by 2x (MI355X_MICROARCH.md, HBM section); hbm_bytes_corrected = '
                '(2*FETCH_SIZE + WRITE_SIZE)*1024 is the corrected upper bound used for roofline.traffic.',
        'kernels': res}


if __name__ == '__main__':
    src, out = sys.argv[1], sys.argv[2]
    res = aggregate(src)
    json.dump(document(res), open(out, 'w'), indent=1)
    print(json.dumps({k: {c: round(v, 1) for c, v in cs.items()} for k, cs in res.items() if k.startswith('k_chain')}, indent=1))
