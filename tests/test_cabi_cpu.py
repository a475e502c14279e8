"""CPU-side checks of the boundary: the shared library loads and exports every symbol include/patchgan_hip.h
declares (no compute calls: there is no GPU here), and the host mirror keeps the reference's surface."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'patchgan_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(pg_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from patchgan_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.SIGNATURES) == names
    assert lib.pg_version() >= 1


def test_workspace_query_and_arg_validation():
    import ctypes
    from patchgan_amd import _lib
    lib = _lib.load()
    g = _lib.ConvGeom(16, 16, 16, 8, 8, 512, 512, 2)
    assert lib.pg_conv_workspace_bytes(ctypes.byref(g), 0) > 0          # small M, long K -> split-K slabs
    bad = _lib.ConvGeom(16, 16, 16, 7, 8, 512, 512, 2)                  # Hs inconsistent with Hb
    assert lib.pg_conv_workspace_bytes(ctypes.byref(bad), 0) == 0
    # argument validation happens before any launch, so it is safe without a GPU
    assert lib.pg_conv4x4_big2small(None, 4, None, None, None, 4, ctypes.byref(g), 0, 0, None, 0, None) == -1
    assert lib.pg_adam_step(None, None, None, None, 10, 1e-3, 0.9, 0.999, 1e-8, 0.1, 0.03, None) == -1
    assert lib.pg_instnorm_act_fwd(None, 4, None, 4, None, 1, 4, 4, 0, 1e-5, 0.0, 0, None, 0, None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from patchgan_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.HipLibraryError):
        _lib.load()


def test_state_dict_surface_matches_oracle_plan():
    import patchgan_amd as pg
    from oracle import patchgan_oracle as O
    g = pg.UNet(3, 2, 4, activation='relu', final_act='sigmoid')
    d = pg.Discriminator(5, 8, n_layers=5, norm=True)
    gs = O.unet_weight_shapes(3, 2, 4)
    ds = O.disc_weight_shapes(5, 8, 5, True)
    assert list(g.state_dict()) == list(gs) and all(tuple(v.shape) == gs[k] for k, v in g.state_dict().items())
    assert list(d.state_dict()) == list(ds) and all(tuple(v.shape) == ds[k] for k, v in d.state_dict().items())
    assert [n for n, _ in g.named_parameters()] == list(gs)
    # parameters are views of the flat packed buffer: writing through state_dict reaches the kernels' memory
    k = 'decoder.3.model.UpConv3.weight'
    w = torch.randn(gs[k])
    g.load_state_dict({**g.state_dict(), k: w})
    lay = [l for l in g.engine.layers if l.key == k][0]
    P = g.flat[lay.p_off:lay.p_off + 16 * lay.a * lay.b].view(4, 4, lay.a, lay.b)
    assert torch.equal(P.permute(2, 3, 0, 1), w)


def test_default_init_equals_torch_module_init():
    """weights_init is a no-op in the reference (trainer.py:327-343): torch's default init must survive, and the
    draw order must match nn.Module construction so a seed reproduces the reference's initial weights."""
    import patchgan_amd as pg
    from tests.golden_util import Golden
    gold = Golden('a_lrelu_tversky')
    torch.manual_seed(gold.model_seed)
    g = pg.UNet(3, 1, 4, activation='leakyrelu', final_act='sigmoid')
    d = pg.Discriminator(4, 4, n_layers=3)
    for k, v in gold.weights('g0').items():
        assert torch.equal(g.state_dict()[k], v), k
    for k, v in gold.weights('d0').items():
        assert torch.equal(d.state_dict()[k], v), k


def test_transfer_and_errors():
    import patchgan_amd as pg
    from patchgan_amd.transfer import InvalidCheckpointError
    g = pg.UNet(3, 1, 4, activation='relu', final_act='sigmoid')
    g2 = pg.UNet(3, 1, 4, activation='relu', final_act='sigmoid')
    g2.load_transfer_data(g.state_dict())
    assert torch.equal(g.flat, g2.flat)
    g3 = pg.UNet(3, 1, 8, activation='relu', final_act='sigmoid')
    sd = {k: v for k, v in g3.state_dict().items()}
    with pytest.raises(InvalidCheckpointError):
        g.load_transfer_data(sd)
    with pytest.raises(ValueError):
        pg.UNet(3, 1, 4, activation='gelu')
