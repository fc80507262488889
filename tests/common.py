"""Helpers shared by the tests: synthetic state dicts keyed like the reference checkpoints."""
import functools

import numpy as np

from avatarcap_amd import config, synthetic as syn
import golden_inputs as gi


def _cfg():
    if not config.cfg:
        config.cfg = config.default_cfg()


@functools.lru_cache(maxsize=None)
def geotex_shapes():
    _cfg()
    from avatarcap_amd.network.arch_avatar import GeoTexAvatar
    net = GeoTexAvatar(base_weight_volume=gi.blend_weight_volume())
    return syn.module_shapes(net)


@functools.lru_cache(maxsize=None)
def geotex_sd(seed=gi.SEED_NET):
    return syn.synth_state_dict(geotex_shapes(), seed)


@functools.lru_cache(maxsize=None)
def recon_sd(seed=gi.SEED_NET):
    from avatarcap_amd.network.arch_recon import ReconNetwork
    return syn.synth_state_dict(syn.module_shapes(ReconNetwork()), seed)


def mlp_sd(name, seed=gi.SEED_MLP):
    from avatarcap_amd.network.mlp import MLP
    m = MLP(**gi.MLP_CONFIGS[name]['kwargs'])
    return syn.synth_state_dict(syn.module_shapes(m), seed)


def offset_decoder_sd(seed=gi.SEED_MLP):
    from avatarcap_amd.network.mlp import OffsetDecoder
    return syn.synth_state_dict(syn.module_shapes(OffsetDecoder(67)), seed)


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))
