"""Shared loaders for the committed golden vectors (tests/golden/, made by make_golden.py)."""
import json
import os

import numpy as np

import joint_cnn_mrf_amd  # noqa: F401
from joint_cnn_mrf_amd import priors, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return np.load(os.path.join(GOLDEN, name + '.npy'))


def stats():
    with open(os.path.join(GOLDEN, 'layer_stats.json')) as fh:
        return json.load(fh)


def seeds():
    return stats()['seeds']


def flic_priors():
    return priors.build_pairwise_distributions(load('flic_train_cells'))


def full_inputs():
    g = seeds()
    p = synth.make_pd_params(debug=False, seed=g['weights'], bn='trained', conv6_gain=g['conv6_gain'])
    return synth.make_images(2, seed=g['images']), synth.make_torso(2, seed=g['torso']), p
