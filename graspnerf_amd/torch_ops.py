"""torch.ops.graspnerf.*: the hot path as registered PyTorch operators (csrc/gnr_torch_ops.cpp, TORCH_LIBRARY).
The operators call the same C-ABI entry points of libgnr.so as the ctypes route (_lib.py / hotpath.py) on the current HIP stream
and return the same bits; they exist so that non-Python hosts, torch.export and the dispatcher see the path.  GPU only: a CPU
tensor makes the dispatcher raise (there is no CPU kernel registered, by design)."""
import os

import torch

from . import _lib

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc', 'libgnr_torch.so')
OPS = ('sample_volume', 'render_rays', 'sample_volume_train', 'sample_volume_bwd')
RENDER_KEYS = ('depth', 'sdf_values', 'alpha_values', 'colors_nr', 'hit_prob_nr', 'pixel_colors_nr', 'pixel_colors_gt', 'render_depth',
               'ray_mask', 'sdf_gradient_error')           # order of the ten tensors render_rays returns per level (coarse, then fine)
_loaded = False


def load():
    """Load libgnr_torch.so once (registers the operators) -> torch.ops.graspnerf.  Raises if it has not been built."""
    global _loaded
    if not _loaded:
        _lib.lib()                                         # libgnr.so first (and torch's ROCm runtime before it, see _lib.lib)
        if not os.path.exists(LIB_PATH):
            raise _lib.GnrError(f'{LIB_PATH} not found: build it with graspnerf_amd/csrc/build.sh')
        torch.ops.load_library(LIB_PATH)
        _loaded = True
    return torch.ops.graspnerf
