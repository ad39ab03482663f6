"""Import the read-only reference (/root/reference) on CPU for golden-vector generation.

Only used in the build container by tools/make_goldens.py; never shipped, never imported
by the product, tests -m gpu, smoke() or bench.py.  The reference hard-codes CUDA placement
(init_net.py:16-17, ibrnet.py:313,444) and imports `easydict` without using it
(aggregate_net.py:4); we patch those three things and nothing else.
"""
import sys, types, torch

REF = '/root/reference'

def import_reference():
    if 'easydict' not in sys.modules:
        m = types.ModuleType('easydict')
        class EasyDict(dict):
            pass
        m.EasyDict = EasyDict
        sys.modules['easydict'] = m
    # Tensor.cuda -> identity ; Tensor.to("cuda:*") -> cpu
    torch.Tensor.cuda = lambda self, *a, **k: self
    _orig_to = torch.Tensor.to
    def _to(self, *a, **k):
        a = tuple(('cpu' if (isinstance(x, str) and x.startswith('cuda')) else x) for x in a)
        if isinstance(k.get('device', None), str) and k['device'].startswith('cuda'):
            k['device'] = 'cpu'
        return _orig_to(self, *a, **k)
    torch.Tensor.to = _to
    for p in (REF + '/src/nr', REF + '/src'):
        if p not in sys.path:
            sys.path.insert(0, p)
    import network.renderer as renderer
    return renderer
