"""tools/pmc_summary.py (shared by tools/collect_profiles.sh and bench.py's live counter passes): per-kernel averages over the
dispatches of the rocprofv3 --pmc CSVs, the per-XCD rows of a dispatch summed, inference kernels of the backward passes dropped, the
gfx950 read correction of the HBM-side bytes (MI355X_MICROARCH.md: FETCH_SIZE under-counts 16 B/lane reads by 2x)."""
import csv, importlib.util, os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mod():
    sp = importlib.util.spec_from_file_location('pmc_summary', os.path.join(ROOT, 'tools', 'pmc_summary.py'))
    m = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(m)
    return m


def _write(path, rows):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'w', newline='') as f:
        w = csv.DictWriter(f, fieldnames=['Dispatch_Id', 'Kernel_Name', 'Counter_Name', 'Counter_Value'])
        w.writeheader()
        for r in rows:
            w.writerow(dict(zip(w.fieldnames, r)))


def test_aggregate_sums_xcd_rows_averages_dispatches_and_corrects_reads(tmp_path):
    pm = _mod()
    chain = 'void gnr::k_chain<6, false, false, false, true>(gnr::ChainArgs)'
    bwd = 'void gnr::k_view1_bwd_pw<false>(gnr::View1BwdArgs)'
    other = 'void at::native::vectorized_elementwise_kernel<4, foo>(int)'
    # forward pass, FETCH_SIZE: two dispatches, each reported as two rows (per-XCD partial sums)
    _write(str(tmp_path / 'FETCH_SIZE' / 'box' / '1_counter_collection.csv'),
           [(1, chain, 'FETCH_SIZE', 100.0), (1, chain, 'FETCH_SIZE', 50.0), (2, chain, 'FETCH_SIZE', 250.0), (3, other, 'FETCH_SIZE', 9e9)])
    _write(str(tmp_path / 'WRITE_SIZE' / 'box' / '1_counter_collection.csv'), [(1, chain, 'WRITE_SIZE', 10.0), (2, chain, 'WRITE_SIZE', 30.0)])
    # backward pass: its inference chain launch (another batch size) must not pollute the forward kernel's average
    _write(str(tmp_path / 'bwd_FETCH_SIZE' / 'box' / '1_counter_collection.csv'),
           [(1, chain, 'FETCH_SIZE', 7777.0), (2, bwd, 'FETCH_SIZE', 400.0),
            (3, 'void gnr::k_chain<6, false, true, false, true>(gnr::ChainArgs)', 'FETCH_SIZE', 60.0)])
    res = pm.aggregate(str(tmp_path))
    k = res['k_chain<6, false, false, false, true>']
    assert k['FETCH_SIZE'] == (150.0 + 250.0) / 2 and k['WRITE_SIZE'] == 20.0
    assert k['hbm_bytes_corrected'] == (2 * 200.0 + 20.0) * 1024
    assert res['k_view1_bwd_pw<false>']['FETCH_SIZE'] == 400.0
    assert res['k_chain<6, false, true, false, true>']['FETCH_SIZE'] == 60.0          # the training forward (SAVE = true) is kept
    assert not any('vectorized' in n for n in res)
    doc = pm.document(res)
    assert len(doc['kernel_source_sha16']) == 16 and len(doc['bwd_source_sha16']) == 16 and doc['kernels'] is res


def test_bench_prefers_counters_of_its_own_run(monkeypatch):
    """bench.newest_pmc: the live document wins; without one the committed file is used only while its stamp matches the sources."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setitem(bench._LIVE_PMC, 'doc', {'kernels': {}, 'kernel_source_sha16': 'x'})
    d, src = bench.newest_pmc('kernel_source_sha16', 'gnr_kernels.hip')
    assert d is bench._LIVE_PMC['doc'] and 'this run' in src
    monkeypatch.setitem(bench._LIVE_PMC, 'doc', None)
    d, src = bench.newest_pmc('kernel_source_sha16', 'gnr_kernels.hip')
    if d is not None:                                          # (the committed file is current in a clean tree)
        assert src.startswith('profiles/') and d['kernel_source_sha16'] == bench.source_sha16('gnr_kernels.hip')
