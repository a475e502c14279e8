"""Transfer-learning partial load (reference patchgan/transfer.py:8-22): copy every tensor of a checkpoint
whose shape matches the live parameter of the same name; raise if nothing matched."""
from torch.nn.parameter import Parameter


class InvalidCheckpointError(Exception):
    pass


class Transferable:
    def load_transfer_data(self, state_dict):
        own = self.state_dict()
        n_loaded = 0
        for name, value in state_dict.items():
            if isinstance(value, Parameter):
                value = value.data
            if value.shape == own[name].data.shape:
                own[name].copy_(value)
                n_loaded += 1
        if n_loaded == 0:
            raise InvalidCheckpointError("Could not load transfer weights")
        print(f"Loaded {n_loaded} weights out of {len(state_dict)}")
