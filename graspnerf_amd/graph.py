"""hipGraph capture of a whole single-scene forward (PyTorch backbones + libgnr.so kernels + grasp head).

Eager PyTorch spends ~30 ms of host time launching the ~1000 small kernels of the 2D backbones for ~9 ms of GPU
work; captured once and replayed, the planner-style forward runs at GPU speed.  All libgnr entry points are
capture-safe: kernel launches on the caller's stream only, no allocation, no host synchronisation.
Requires cfg['depth_coords_rng'] = 'device' (a CPU randperm cannot be captured)."""
import torch


class GraphedForward:
    def __init__(self, net, example_data, warmup=3):
        if net.nr_net.cfg.get('depth_coords_rng', 'cpu') != 'device':
            raise ValueError("graph capture needs cfg['depth_coords_rng'] = 'device'")
        self.net = net
        self.static = self._clone(example_data)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(warmup):                      # builds HotPath / GraspHead, sets kernel attributes
                net(self.static)
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.out = net(self.static)

    @staticmethod
    def _clone(d):
        return {k: (GraphedForward._clone(v) if isinstance(v, dict) else (v.clone() if torch.is_tensor(v) else v))
                for k, v in d.items()}

    def _copy(self, dst, src):
        for k, v in src.items():
            if isinstance(v, dict):
                self._copy(dst[k], v)
            elif torch.is_tensor(v):
                dst[k].copy_(v)

    def __call__(self, data):
        self._copy(self.static, data)
        self.graph.replay()
        return self.out
