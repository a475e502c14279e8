"""Trainer.batch returns StepLosses: a dict of the six loss scalars (reference trainer.py:108-115) whose values arrive with the
step's asynchronous device-to-host copy.  Host-only check of the dict surface with a stand-in event."""
import copy
import json
import pickle

import numpy as np
import torch


class _Event:
    def __init__(self):
        self.waited = 0

    def synchronize(self):
        self.waited += 1


def test_step_losses_behaves_like_the_reference_dict():
    from patchgan_amd.trainer import StepLosses
    host = torch.tensor([3.0, 0.5, 0.25, 0.125, 0, 0, 0, 0])
    ev = _Event()
    l = StepLosses(host, ev)
    assert ev.waited == 0                       # nothing read yet: nothing waited for
    assert l['gen'] == 3.5 and ev.waited == 1   # seg + gdisc
    want = {'gen': 3.5, 'gen_loss': 3.5, 'gdisc': 0.5, 'discr': 0.5, 'discf': 0.25, 'disc': 0.375}
    assert l == want and dict(l) == want and list(l) == list(want) and len(l) == 6 and 'disc' in l
    assert {k: v for k, v in l.items()} == want and list(l.values()) == list(want.values()) and l.get('nope', 1) == 1
    assert json.loads(json.dumps(l)) == want and pickle.loads(pickle.dumps(l)) == want and copy.copy(l) == want
    assert ev.waited == 1                       # one wait, however often it is read
    host[0] = 100.0                             # the pinned slot is reused by a later step: the values were taken out
    assert l['gen'] == 3.5
    a, b = StepLosses(torch.ones(8), _Event()), StepLosses(torch.ones(8), _Event())
    assert [a] == [b] and not (a != b)          # two unread results compare by value
    assert isinstance(l, dict) and all(isinstance(v, float) for v in l.values())
    assert np.isclose(sum(l.values()), 8.625)
