"""Atom-count prior: Categorical over the training-set histogram of molecule sizes.
Behaviour of /root/reference/models/node_distribution.py:5-48; histograms are the data in
jodo_amd/data/n_nodes_hist.json (extracted by tools/extract_histograms.py)."""
import json
import os

import torch
from torch.distributions.categorical import Categorical

_HIST_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'data', 'n_nodes_hist.json')


def load_dataset_info(info_name):
    """info_name in {'qm9_with_h', 'qm9_second_half', 'geom_with_h_1'} -> {'train_n_nodes', 'max_n_nodes'}"""
    with open(_HIST_FILE) as f:
        raw = json.load(f)[info_name]
    return {'max_n_nodes': raw['max_n_nodes'],
            'train_n_nodes': {int(k): v for k, v in raw['train_n_nodes'].items()}}


class DistributionNodes:
    def __init__(self, histogram, verbose=False):
        sizes = list(histogram.keys())
        self.keys = {n: i for i, n in enumerate(sizes)}
        self.n_nodes = torch.tensor(sizes)
        prob = torch.tensor([histogram[n] for n in sizes])
        self.prob = prob / torch.sum(prob)
        if verbose:
            ent = torch.sum(self.prob * torch.log(self.prob + 1e-30))
            print("Entropy of n_nodes: H[N]", ent.item())
        self.m = Categorical(self.prob)

    def sample(self, n_samples=1):
        return self.n_nodes[self.m.sample((n_samples,))]

    def log_prob(self, batch_n_nodes):
        assert batch_n_nodes.dim() == 1
        idx = torch.tensor([self.keys[i.item()] for i in batch_n_nodes], device=batch_n_nodes.device)
        return torch.log(self.prob + 1e-30).to(batch_n_nodes.device)[idx]


def get_node_dist(dataset_info):
    return DistributionNodes(dataset_info['train_n_nodes'])
