"""Checkpoint format of the reference (utils/pytorch_utils.py:7-51): a dict
``{'states': net.state_dict(), **kwargs}`` saved with torch.save; extra keys in the file are dropped
on load and the training step is returned."""
import os
from collections import OrderedDict

import torch


def save_network(net, directory, network_label, epoch_label=None, **kwargs):
    """(reference :7-15)"""
    save_filename = "_".join((network_label, str(epoch_label))) + ".pth"
    if not os.path.exists(directory):
        os.makedirs(directory)
    merge_states = OrderedDict()
    merge_states['states'] = OrderedDict((k, v.detach().cpu()) for k, v in net.state_dict().items())
    for k in kwargs:
        merge_states[k] = kwargs[k]
    path = os.path.join(directory, save_filename)
    torch.save(merge_states, path)
    return path


def load_network(net, path):
    """load the parameters whose names exist in `net`; return the trained step (reference :18-51)"""
    loaded_state = torch.load(path, map_location="cpu")
    network = net.module if isinstance(net, torch.nn.DataParallel) else net
    own_state = network.state_dict()
    extra = set(loaded_state["states"].keys()) - set(own_state.keys())
    if len(extra) > 0:
        print('Dropping ' + str(extra) + ' from loaded states')
    for k in extra:
        del loaded_state["states"][k]
    try:
        network.load_state_dict(loaded_state["states"])
    except KeyError as e:
        print(e)
        return 0
    print('Loaded network parameters from {}'.format(path))
    return int(loaded_state["step"]) if "step" in loaded_state else 0
