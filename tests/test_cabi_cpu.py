"""CPU-side checks of the boundary: the shared library loads and exports every symbol include/patchgan_hip.h
declares (no compute calls: there is no GPU here), and the host mirror keeps the reference's surface."""
import ctypes
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


def test_kernel_plan_of_the_benchmark_layers():
    """pg_conv_describe is host-only: which kernel family PG_ALGO_AUTO / PG_ALGO_MFMA pick for the cfg2 layer shapes
    (bs 16), and that the workspace query covers the Winograd paths."""
    import ctypes
    from patchgan_amd import engine as E, _lib
    if os.environ.get('PATCHGAN_WINO2') or os.environ.get('PATCHGAN_NO_WINOGRAD'):
        pytest.skip('kernel-selection switches set in the environment')
    want = {   # (N, Hb, Wb, Ca, Cb, stride): (big2small, small2big, wgrad) kernel families under PG_ALGO_AUTO
        (16, 256, 256, 64, 3, 2): ('k_b2s_tapk', 'k_s2b_tapnf<3>', 'k_wgrad_tapn'),     # enc0
        (16, 128, 128, 128, 64, 2): ('k_wino_bgemm_s3<2,2,2,2,2>', 'k_wino_bgemm_s3<2,2,2,2,2>', 'k_wino_wgrad_gemm_s3<2,2,2,2,1,2>'),   # enc1
        (16, 64, 64, 256, 128, 2): ('k_wino_bgemm_s3<2,2,2,2,2>', 'k_wino_bgemm_s3<2,2,2,2,2>', 'k_wino_wgrad_gemm_s3<2,2,2,2,1,2>'),    # enc2
        (16, 16, 16, 512, 512, 2): ('k_b2s_fast', 'k_s2b_fast', 'k_wgrad_fast'),                              # enc4: too few tiles
        (16, 64, 64, 512, 128, 2): ('k_wino_bgemm_s3<2,2,2,2,2>', 'k_wino_bgemm_s3<2,2,2,2,2>', 'k_wino_wgrad_gemm_s3<2,2,2,2,1,2>'),    # dec4
        (16, 32, 32, 1024, 256, 2): ('k_wino_bgemm_s3<1,2,2,2,3>', 'k_wino_bgemm_s3<1,2,2,2,3>', 'k_wino_wgrad_gemm_s3<2,2,2,2,1,2>'),   # dec3
        (32, 32, 32, 512, 256, 1): ('k_wino_gemm', 'k_wino_gemm', 'k_wino_wgrad_gemm'),                       # d3 at 2N
        (32, 31, 31, 1, 512, 1): ('k_b2s_fast<1,1,4,1,true>+k_gather', 'k_s2b_ca1', 'k_wgrad_tapn'),        # D head
    }
    for geom, fams in want.items():
        auto, mfma = E.ConvOp(*geom, _lib.ALGO_AUTO), E.ConvOp(*geom, _lib.ALGO_MFMA)
        for oc, fam in enumerate(fams):
            sym = auto.describe(oc)[0]
            assert sym.startswith(fam), (geom, oc, sym)
            assert 'wino' not in mfma.describe(oc)[0], (geom, oc)
            if '_s3<' in sym:      # PG_TUNE_S3_OFF: the same plan on the fp32 MFMA
                off = E.ConvOp(*geom, _lib.ALGO_AUTO | _lib.TUNE_S3_OFF).describe(oc)[0]
                if oc == 2:            # (the fp32 weight-gradient GEMM keeps its own tile choice: 64x64 tiles on most layers)
                    assert off.startswith('k_wino_wgrad_gemm<'), (sym, off)
                elif sym.startswith('k_wino_gemm_row_s3'):
                    assert off == 'k_wino_gemm_row<4,1>', (sym, off)
                else:
                    assert off == sym.replace('_s3<', '<')[:-3] + '>', (sym, off)
            assert auto.kernel_flops(oc) <= auto.flops * 1.3       # Winograd executes fewer (ragged tiles may add a little)
            if 'wino' in sym:
                assert auto.kernel_flops(oc) < 0.6 * auto.flops
        g = _lib.ConvGeom(geom[0], geom[1], geom[2], auto.Hs, auto.Ws, geom[3], geom[4], geom[5])
        assert auto.ws_bytes == max(_lib.load().pg_conv_workspace_bytes(ctypes.byref(g), oc) for oc in range(4))   # 3 = pg_conv4x4_bwd_big


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


def test_queries_do_not_depend_on_a_larger_workspace():
    """The engine sizes hand-over buffers from queries (pg_conv_kernel, pg_conv_u_bytes, pg_conv_v_bytes, pg_conv_stats_chunks,
    pg_conv_mul_ok) and then launches; both sides must see the same plan.  ConvOp passes ONE workspace size to both
    (ConvOp.ws_arg), and -- checked here on every layer geometry of cfg2, cfg4 and cfg1 -- the C side's answers do not change
    with any workspace at least as large as pg_conv_workspace_bytes reports (the device buffer the launches share may be larger)."""
    import patchgan_amd as pg
    from patchgan_amd import _lib, engine as E
    lib = _lib.load()
    geoms = set()
    for nf, ndf, nl, out_nc, size, B in ((64, 64, 3, 1, 256, 16), (64, 64, 3, 4, 512, 8), (32, 16, 5, 7, 256, 4)):
        g, d = E.GeneratorEngine(3, out_nc, nf, 'leakyrelu', 'sigmoid', False), E.DiscriminatorEngine(3 + out_nc, ndf, nl, False)
        enc, dec = g.ops(B, size, size)
        for op in enc + dec + d.ops(B, size, size) + d.ops(2 * B, size, size):
            geoms.add(op.g.key())
    assert len(geoms) > 40
    for key in sorted(geoms):
        for algo, ios in ((_lib.ALGO_AUTO, (0,)), (_lib.ALGO_BF16, (0, _lib.IO_MASK))):
            op = E.ConvOp(key[0], key[1], key[2], key[5], key[6], key[7], algo)
            assert op._ws(torch.device('cpu'))[1] == op.ws_arg       # what a launch passes == what the queries pass
            for io in ios:
                a = algo | io

                def answers(ws):
                    out = []
                    for oc in (0, 1, 2):
                        name, s, fl = ctypes.create_string_buffer(128), ctypes.c_int(0), ctypes.c_double(0)
                        assert lib.pg_conv_kernel(ctypes.byref(op.g), oc + 16 * a, ws, name, 128, ctypes.byref(s), ctypes.byref(fl)) == 0
                        out.append((name.value, s.value, fl.value))
                    out += [lib.pg_conv_u_bytes(ctypes.byref(op.g), oc, a, ws) for oc in (0, 1)]
                    out += [lib.pg_conv_stats_chunks(ctypes.byref(op.g), oc, a, ws) for oc in (0, 1)]
                    out += [lib.pg_conv_v_bytes(ctypes.byref(op.g), a, ws), lib.pg_conv_mul_ok(ctypes.byref(op.g), a, ws)]
                    return out
                base = answers(op.ws_arg)
                for bigger in (op.ws_arg + 4096, 4 * op.ws_arg, 1 << 33):
                    assert answers(bigger) == base, (key, algo, io, bigger)


def test_weight_prep_pools_are_bounded_and_cleared():
    """engine._WeightPrep: ONE device-buffer pool per network keyed by entry (not per input extent), at most MAX_PLANS remembered
    fill plans, both dropped by set_precision / set_tuning."""
    import patchgan_amd as pg
    from patchgan_amd import engine as E
    g = pg.UNet(3, 1, 4, activation='relu', final_act='sigmoid')
    eng = g.engine
    for i in range(E._WeightPrep.MAX_PLANS + 5):
        uc = E.UCache(eng.__dict__.setdefault('_upool', {}))
        uc.plan_key = (2 + i, 256, 256, None, eng.algo, False)
        buf = uc[('e', 1, 4096)] = uc.buffer(('e', 1, 4096), 4096, torch.device('cpu'))       # the same entry at every extent
        uc.log.append((('e', 1, 4096), None, 0, 0, 0, 4096))
        if i == 3:                                                                         # an entry only one (evicted) plan used
            uc[('x', 0, 64)] = uc.buffer(('x', 0, 64), 64, torch.device('cpu'))
            uc.log.append((('x', 0, 64), None, 0, 0, 0, 64))
        eng.ucache_end(uc)
        assert eng._upool[('e', 1, 4096)] is buf                                             # one buffer, reused by every extent
    assert len(eng._uplan) == E._WeightPrep.MAX_PLANS
    assert list(eng._upool) == [('e', 1, 4096)]                                              # the evicted plan's private entry is gone
    g.set_tuning(0)
    assert '_upool' not in eng.__dict__ and '_uplan' not in eng.__dict__
    eng.__dict__['_upool'] = {1: 2}
    g.set_precision('fp32')
    assert '_upool' not in eng.__dict__


def test_tensors_beyond_the_32_bit_offset_limit_are_never_planned_onto_the_bf16_kernels():
    """A bf16 tensor of pg_conv_max_tensor_bytes (1.5 GiB) or more has no kernel (the buffer-load kernels address with 32-bit byte
    offsets) and the C ABI's size queries see the geometry only.  The engine therefore (a) keeps fp32 activation storage for an
    input extent any of whose interior tensors -- as the channel slice of the wider skip buffer it lives in -- would reach the
    limit, and (b) plans no hand-over (InstanceNorm partials, packed-weight cache, V hand-over, epilogue multiplier) for such a view.
    No launch: queries and host logic only."""
    from patchgan_amd import _lib, engine as E
    limit = _lib.load().pg_conv_max_tensor_bytes()
    assert limit == 0x60000000
    g = E.GeneratorEngine(3, 1, 64, 'leakyrelu', 'sigmoid', False, algo=_lib.ALGO_BF16)
    g.act_bf = True
    for N, ok in ((16, True), (383, True), (384, False), (400, False)):   # enc0's output in cat6: N * 128*128 * 128 ch * 2 B = N * 4 MiB
        enc, dec = g.ops(N, 256, 256)
        assert g._storage_ok((N, 256, 256), enc[1:] + dec[:6]) == ok, N
    d = E.DiscriminatorEngine(4, 64, 3, False, algo=_lib.ALGO_BF16)
    for N, ok in ((32, True), (767, True), (768, False)):                 # d0's output: N * 128*128 * 64 ch * 2 B = N * 2 MiB
        ops = d.ops(N, 256, 256)
        assert d._storage_ok((N, 256, 256), ops[1:-1]) == ok, N

    class FakeTensor:          # a device address without device memory
        device = torch.device('cpu')

        def data_ptr(self):
            return 1 << 21

    def view(N, H, W, C, ld):
        return E.View(FakeTensor(), 0, ld, N, H, W, C, True)
    for N, fits in ((16, True), (400, False)):
        op = E.ConvOp(N, 128, 128, 128, 64, 2, _lib.ALGO_BF16)            # enc1 on bf16 tensors, its input a slice of cat6 (ld 128)
        src, dst = view(N, 128, 128, 64, 128), view(N, 64, 64, 128, 256)
        assert E.ConvOp.fits(src, dst) == fits
        assert op.u_bytes(0, _lib.IO_MASK) > 0                            # the geometry-only query always offers a packed-weight cache
        uc = E.UCache()
        buf, valid = E._ucache(uc, ('e', 1), 0, op, torch.device('cpu'), src, dst, 0)
        assert (buf is not None) == fits and not valid
        assert bool(op.stats_chunks(0, src, dst)) == (fits and bool(_lib.load().pg_conv_stats_chunks(
            ctypes.byref(op.g), 0, _lib.ALGO_BF16 | _lib.IO_MASK, op.ws_arg)))
        t = view(N, 128, 128, 64, 128)
        assert op.mul_ok(dst, view(N, 128, 128, 64, 128), t) in ((True, False) if fits else (False,))


def test_bench_kernel_plan_is_the_committed_one():
    """tests/golden/bench_kernel_plan.json (written by tools/dump_kernel_plan.py) names, for every conv call of the benchmark
    configurations at their exact batch sizes, the kernel the planner picks; the per-layer GPU parity test carries those symbols in
    its ids.  A planner or kernel change that moves a call to another kernel fails HERE, on the CPU, until the plan is re-dumped
    (and the new kernel thereby enters the per-layer test)."""
    from tests import bench_layers as BL
    live, committed = BL.live_plan(), BL.committed_plan()
    assert sorted(live) == sorted(committed)
    diff = {k: (committed[k], live[k]) for k in live if live[k] != committed[k]}
    assert not diff, diff


def test_one_pass_transposed_conv_onto_few_channels_is_planned_where_it_exists():
    """Stride-2 small -> big onto <= 8 channels from 32 / 64 / 128 channels runs the one-pass kernel k_s2b_tapnf (D block in LDS; 5-8
    channels as two launches of <= 4); other channel counts and stride 1 keep the two-launch row GEMM + col2im."""
    from patchgan_amd import engine as E, _lib as L
    for Ca in (32, 64, 128):
        for Cb in (1, 2, 3, 4):
            assert E.ConvOp(2, 32, 32, Ca, Cb, 2, L.ALGO_AUTO).describe(1)[0] == f'k_s2b_tapnf<{Cb}>'
    assert E.ConvOp(2, 32, 32, 64, 7, 2, L.ALGO_AUTO).describe(1)[0] == 'k_s2b_tapnf<4>+k_s2b_tapnf<3>'
    for geom in ((2, 32, 32, 96, 2, 2), (2, 32, 32, 256, 6, 2), (2, 9, 9, 32, 4, 1)):
        assert E.ConvOp(*geom, L.ALGO_AUTO).describe(1)[0].endswith('k_col2im_small2big'), geom
    # a bf16 `small` (fp32 result): the same kernel on bf16 MFMAs; both tensors bf16: the bf16 kernels of the wide layers' family
    assert E.ConvOp(2, 32, 32, 64, 4, 2, L.ALGO_BF16).describe(1, L.IO_SMALL_BF16)[0] == 'k_s2b_tapnf<4,bf16>'
    assert E.ConvOp(2, 32, 32, 64, 7, 2, L.ALGO_BF16).describe(1, L.IO_SMALL_BF16)[0] == 'k_s2b_tapnf<4,bf16>+k_s2b_tapnf<3,bf16>'
    assert 'tapnf' not in E.ConvOp(2, 32, 32, 64, 4, 2, L.ALGO_BF16).describe(1, L.IO_SMALL_BF16 | L.IO_BIG_BF16)[0]


def test_useful_flops_are_the_executed_ones_without_tile_padding():
    """pg_conv_kernel_flops: executed >= useful, equal where no ragged tiles exist; the polyphase F(2x2,3x3) weight gradient of a
    16 x 16 map (cfg2 enc3: 6 x 6 tiles of 3 for 16 rows) executes (18 / 16)^2 = 1.27x its useful count."""
    from patchgan_amd import engine as E
    op = E.ConvOp(16, 32, 32, 512, 256, 2, E.DEFAULT_ALGO)          # enc3 at cfg2
    name, _ = op.describe(2)
    ex, us = op.kernel_flops(2), op.useful_flops(2)
    if name.startswith('k_wino_wgrad_gemm'):
        assert abs(ex / us - (18 / 16) ** 2) < 1e-9, (ex, us)
        assert us < op.flops                                        # Winograd: fewer multiplies than the direct count
    for oc in (0, 1, 2):
        assert op.useful_flops(oc) <= op.kernel_flops(oc) * (1 + 1e-12)
    small = E.ConvOp(16, 8, 8, 512, 512, 2, E.DEFAULT_ALGO)         # enc5: implicit GEMM, no padding to report
    for oc in (0, 1, 2):
        assert small.useful_flops(oc) == small.kernel_flops(oc) == small.flops
