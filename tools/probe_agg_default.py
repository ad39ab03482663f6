"""What does the REFERENCE do with agg_net_type 'default' (the density branch: renderer.py:21,98-100, aggregate_net.py:72-85,
ibrnet.py:240-371)?  Builds NeuralRayRenderer / GraspNeRF from the imported reference (tools/ref_import.py, this container only) with the
reference's yaml and agg_net_type switched, random-initialised, and calls sample_volume-free render on a cfg1 scene; records whether the
call runs and, if not, the exception and where it is raised -> tests/golden/ref_agg_default_probe.json (data, read by
tests/test_model_mirror_config.py)."""
import json, os, sys, traceback
import numpy as np, torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from ref_import import import_reference, REF
from graspnerf_amd.synth import make_scene

renderer = import_reference()
out = {}
for name, over in (('agg_net_type default, use_sdf from the yaml', {}), ('agg_net_type default, volume_type alpha', {'volume_type': ['alpha']})):
    cfg = yaml.load(open(REF + '/src/nr/configs/nrvgn_sdf.yaml'), Loader=yaml.FullLoader)
    for k, v in (('volume_resolution', 16), ('depth_sample_num', 16), ('fine_depth_sample_num', 16), ('use_hierarchical_sampling', False), ('agg_net_type', 'default')):
        cfg[k] = v
    cfg.update(over)
    cfg['agg_net_cfg']['sample_num'] = 16
    cfg['fine_agg_net_cfg']['sample_num'] = 16
    try:
        torch.manual_seed(0)
        nr = renderer.NeuralRayRenderer(cfg).eval()
        ref, que = make_scene(0, 'cfg1')
        t = lambda a: torch.from_numpy(a.copy())
        ref_info = {k: t(v) for k, v in ref.items() if k not in ('img_feats', 'ray_feats')}
        que_info = {'coords': t(que['coords'])[None], 'poses': t(que['pose'])[None], 'Ks': t(que['K'])[None], 'depth_range': t(que['depth_range'])[None]}
        with torch.no_grad():
            ri = dict(ref_info)
            ri['img_feats'] = nr.image_encoder(ri['imgs'])
            ri['ray_feats'] = nr.vis_encoder(nr.init_net(ri, None, False), ri['img_feats'])
            o = nr.render(que_info, ri, False)
        out[name] = {'runs': True, 'keys': sorted(o)}
    except Exception as e:
        tb = traceback.extract_tb(e.__traceback__)
        out[name] = {'runs': False, 'exception': type(e).__name__ + ': ' + str(e)[:240],
                     'raised_at': [f'{os.path.relpath(f.filename, REF)}:{f.lineno} {f.name}' for f in tb if f.filename.startswith(REF)][-3:]}
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(ROOT, 'tests/golden/ref_agg_default_probe.json'), 'w'), indent=1)
