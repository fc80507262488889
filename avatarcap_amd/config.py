"""Process-global configuration, same two-layer surface as the reference's `config.py:1-31`:
(1) module globals (`device`, `smpl_gender`, `N_samples`, `perturb`, `if_type` -> `iso_value`,
`sdf_thres`), (2) the yaml dict `cfg` filled by `load_config` (keys of configs/example.yaml).
"""
import torch

# reference: torch.device('cuda') unconditionally (config.py:3); on ROCm 'cuda' is the HIP device.
device = torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')

smpl_gender = 'M'      # config.py:6
N_samples = 64         # config.py:9
perturb = 1            # config.py:10
if_type = 'sdf'        # config.py:13  ('sdf' | 'occupancy')


def _iso_for(t):
    if t == 'sdf':
        return 0.
    if t == 'occupancy':
        return 0.5
    raise ValueError('Invalid if_type!')   # config.py:22


iso_value = _iso_for(if_type)
sdf_thres = 0.1

cfg = dict()           # configurations from the yaml file (config.py:25)

# not in the reference: opt-in check that no feature / activation leaves the fp16 range of the fused queries (include/avcap.h,
# 'numeric range'); a query then synchronises and raises AvcapError (AVC_ERR_RANGE) instead of returning silently wrong values
check_range = False

# not in the reference: the HGFilter encoder and the warping field's U-Net (csrc/conv_enc.hip) replay their ~70 / 18 launches as one hipGraph each per
# frame (avc_set_option "enc_graph"); False launches the same kernels one by one -- same bits
hg_graph = True


def load_config(path):
    import yaml
    with open(path, encoding='UTF-8') as f:
        return yaml.load(f, Loader=yaml.FullLoader)


def default_cfg():
    """The `model.*` / `testing.*` keys the hot path reads, with configs/example.yaml's values."""
    return {
        'training': {'training_data_dir': None},
        'testing': {'vol_res': [384, 384, 128], 'recon_net_ckpt': None, 'net_ckpt': None,
                    'net_ckpt_finetuned': None, 'testing_data_dir': None, 'output_dir': './results/example/testing'},
        'model': {'cano_template': {'pos_encoding': 10}, 'warping_field': {'pos_encoding': 0}},
    }
