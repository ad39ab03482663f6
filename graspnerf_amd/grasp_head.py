"""HIP grasp head: gd.networks.ConvNet.forward on fp32 implicit-GEMM MFMA kernels (csrc/gnr_head.hip).
State-dict keys are the reference's (`vgn_net.encoder.conv1.weight`, ... ; ref: src/gd/networks.py:39-47)."""
import ctypes as C

import numpy as np
import torch

from . import _lib

HEAD_KEYS = [('encoder.conv1', (16, 1, 5)), ('encoder.conv2', (32, 16, 3)), ('encoder.conv3', (64, 32, 3)),
             ('decoder.conv1', (64, 64, 3)), ('decoder.conv2', (32, 64, 3)), ('decoder.conv3', (16, 32, 5)),
             ('conv_qual', (1, 16, 5)), ('conv_rot', (4, 16, 5)), ('conv_width', (1, 16, 5))]


def canonical_blob(state_dict, prefix=''):
    parts = []
    for name, (co, ci, k) in HEAD_KEYS:
        w, b = state_dict[prefix + name + '.weight'], state_dict[prefix + name + '.bias']
        w = w.detach().cpu().numpy() if hasattr(w, 'detach') else np.asarray(w)
        b = b.detach().cpu().numpy() if hasattr(b, 'detach') else np.asarray(b)
        if tuple(w.shape) != (co, ci, k, k, k) or tuple(b.shape) != (co,):
            raise ValueError(f'{name}: unexpected shape {tuple(w.shape)}')
        parts += [np.asarray(w, np.float32).reshape(-1), np.asarray(b, np.float32).reshape(-1)]
    blob = np.concatenate(parts)
    assert blob.size == _lib.lib().gnr_head_canonical_floats()
    return blob


def pack(canonical):
    L = _lib.lib()
    canonical = np.ascontiguousarray(canonical, np.float32)
    out = np.zeros(L.gnr_head_packed_floats(), np.float32)
    rc = L.gnr_pack_grasp_head(canonical.ctypes.data_as(_lib.c_float_p), out.ctypes.data_as(_lib.c_float_p))
    if rc:
        raise _lib.GnrError(f'gnr_pack_grasp_head failed: {rc}')
    return out


class GraspHead:
    def __init__(self, state_dict, prefix='', device='cuda:0'):
        self.L = _lib.lib()
        if not torch.cuda.is_available():
            raise _lib.GnrError('the HIP grasp head needs a ROCm GPU; there is no CPU fallback')
        self.device = torch.device(device)
        self.w = torch.from_numpy(pack(canonical_blob(state_dict, prefix))).to(self.device)
        self._ws = None

    def __call__(self, volume):
        """volume [B,1,R,R,R] (cuda, fp32) -> (qual [B,1,40,40,40], rot [B,4,40,40,40], width [B,1,40,40,40])"""
        vol = volume.to(device=self.device, dtype=torch.float32).contiguous()
        B, _, R = vol.shape[:3]
        need = self.L.gnr_grasp_head_workspace_bytes(B, R)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        q = torch.empty(B, 1, 40, 40, 40, device=self.device)
        r = torch.empty(B, 4, 40, 40, 40, device=self.device)
        w = torch.empty(B, 1, 40, 40, 40, device=self.device)
        rc = self.L.gnr_grasp_head_fwd(B, R, vol.data_ptr(), self.w.data_ptr(), q.data_ptr(), r.data_ptr(), w.data_ptr(),
                                       self._ws.data_ptr(), self._ws.numel(),
                                       C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        if rc:
            raise _lib.GnrError(f'gnr_grasp_head_fwd failed: {_lib.ERRORS.get(rc, rc)} '
                                f'({self.L.gnr_head_last_error().decode(errors="replace")})')
        return q, r, w
