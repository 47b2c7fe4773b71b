"""Checkpoint files in the reference's format (utils/pytorch_utils.py:7-51): one dict, the network's state_dict under
'states' next to whatever bookkeeping the caller adds ('step', ...), written with torch.save as
<label>_<epoch>.pth -- so that the reference's published checkpoints load here and files written here load there.
A checkpoint may also be a pickled numpy dict of the same layout (any extension but .pth; reference :24-27)."""
import os
from collections import OrderedDict

import numpy as np
import torch


def save_network(net, directory, network_label, epoch_label=None, **kwargs):
    """Write <directory>/<network_label>_<epoch_label>.pth (reference :7-15).  The tensors are copied to the host one
    by one; the network itself stays on its device (the reference moves it to the CPU and back)."""
    os.makedirs(directory, exist_ok=True)
    record = OrderedDict(states=OrderedDict((name, t.detach().cpu()) for name, t in net.state_dict().items()))
    record.update(kwargs)
    target = os.path.join(directory, "%s_%s.pth" % (network_label, epoch_label))
    torch.save(record, target)
    return target


def _read_record(path):
    if path.endswith("pth"):
        return torch.load(path, map_location="cpu")
    record = np.load(path, allow_pickle=True).item()            # a 0-d object array holding the dict
    record["states"] = OrderedDict(
        (name, value if torch.is_tensor(value) else torch.from_numpy(np.asarray(value)))
        for name, value in record["states"].items())
    return record


def load_network(net, path):
    """Copy into `net` every parameter of the file that `net` has a slot for; entries of the file the network does
    not know are reported and ignored.  Returns the training step stored in the file, 0 if there is none or if the
    file does not fit the network (reference :18-51)."""
    record = _read_record(path)
    target = net.module if isinstance(net, torch.nn.DataParallel) else net
    known = target.state_dict().keys()
    unknown = [name for name in record["states"] if name not in known]
    if unknown:
        print("Dropping " + str(set(unknown)) + " from loaded states")
    weights = OrderedDict((name, t) for name, t in record["states"].items() if name in known)
    try:
        target.load_state_dict(weights)
    except KeyError as err:
        print(err)
        return 0
    print("Loaded network parameters from {}".format(path))
    return int(record.get("step", 0))
