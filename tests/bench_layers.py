"""The conv calls the benchmark really times, one by one: every distinct layer of the generator and the discriminator at the
EXACT geometry and batch size of bench.py's configurations (cfg2: 256x256, 16 per GPU, D at 16 and 32; cfg4: 512x512, 8 per GPU,
4 classes, D at 8 and 16), for fp32 tensors (PG_ALGO_AUTO, the default planner) and for bf16 activation storage (PG_ALGO_BF16),
with the storage type and pixel stride each operand has inside the engines (skip-connection buffers: ld = 2 C; image-facing
tensors in 8-channel bf16 pixels).

Host-only (the planner is queried through pg_conv_kernel, no launch): tests/test_bench_layers_gpu.py runs every case against
torch float64 on the GPU, tests/test_cabi_cpu.py holds the kernel symbols to tests/golden/bench_kernel_plan.json
(tools/dump_kernel_plan.py writes it), and tests/test_bench_gpu.py checks that every symbol bench.py reports is in that plan."""
import json
import os

CONFIGS = {
    'cfg2': dict(size=256, batch=16, out_nc=1, nf=64, ndf=64, n_layers=3, activation='leakyrelu', final_act='sigmoid'),
    'cfg4': dict(size=512, batch=8, out_nc=4, nf=64, ndf=64, n_layers=3, activation='leakyrelu', final_act='softmax'),
}
MODES = (('cfg2', 'fp32'), ('cfg4', 'fp32'), ('cfg4', 'bf16'), ('cfg2', 'bf16'))
PLAN_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'bench_kernel_plan.json')
OPCODES = {'b2s': 0, 's2b': 1, 'wgrad': 2}


class Operand:
    """Storage of one activation operand of a call: element type, pixel stride as a multiple of its channel count, 8-channel
    bf16 pixels (image-facing tensors of the bf16 mode)."""
    __slots__ = ('bf', 'ldm', 'pad8', 'ld', 'off')

    def __init__(self, bf=False, ldm=1, pad8=False, ld=None, off=None):
        self.bf, self.ldm, self.pad8 = bf, ldm, pad8
        self.ld, self.off = ld, off            # absolute pixel stride / channel offset (the image-facing slices of din), else derived


class Case:
    """One conv call of the step: layer name, geometry, op ('b2s' | 's2b' | 'wgrad' | 'bwd_big'), operand storage, epilogue."""

    def __init__(self, cfg, mode, layer, geom, op, big, small, bias=False, act='none', role=''):
        self.cfg, self.mode, self.layer, self.geom, self.op = cfg, mode, layer, geom, op
        self.big, self.small, self.bias, self.act, self.role = big, small, bias, act, role

    @property
    def io(self):
        from patchgan_amd import _lib as L
        return (L.IO_BIG_BF16 if self.big.bf else 0) | (L.IO_SMALL_BF16 if self.small.bf else 0)

    @property
    def algo(self):
        from patchgan_amd import _lib as L
        return L.ALGO_BF16 if self.mode == 'bf16' else L.ALGO_AUTO

    def convop(self):
        from patchgan_amd import engine as E
        return E.ConvOp(*self.geom, self.algo)

    def symbols(self):
        """Kernel symbol(s) of the main GEMM kernel(s) of this call, as bench.py's conv_kernels table names them."""
        op = self.convop()
        if self.op == 'bwd_big':
            return [op.describe(2, self.io)[0], op.describe(0, self.io)[0]]
        return [op.describe(OPCODES[self.op], self.io)[0]]

    @property
    def key(self):
        return f'{self.cfg}-{self.mode}-{self.layer}-{self.op}' + (f'-{self.role}' if self.role else '')


def cases(cfg, mode):
    """Every conv call of one G+D training step of `cfg` in `mode` ('fp32' | 'bf16'), deduplicated by (geometry, op, storage)."""
    from patchgan_amd import engine as E, _lib as L
    c = CONFIGS[cfg]
    B, S = c['batch'], c['size']
    bfm = mode == 'bf16'
    algo = L.ALGO_BF16 if bfm else L.ALGO_AUTO
    ge = E.GeneratorEngine(3, c['out_nc'], c['nf'], c['activation'], c['final_act'], False, algo=algo)
    de = E.DiscriminatorEngine(3 + c['out_nc'], c['ndf'], c['n_layers'], False, algo=algo)
    out = []

    def geom(op):
        return (op.N, op.Hb, op.Wb, op.Ca, op.Cb, op.stride)

    enc_ops, dec_ops = ge.ops(B, S, S)
    I = Operand
    for i, op in enumerate(enc_ops):                     # Conv2d -> InstanceNorm -> act (unet.py:19-30)
        # enc0 reads x as the first channels of the discriminator-input buffer (ld = 3 + out_nc); enc_i the second half of cat_{7-i}
        big = I(bfm, 1, pad8=True) if (i == 0 and bfm) else I(False, ld=3 + c['out_nc'], off=0) if i == 0 else I(bfm, 2)
        small = I(bfm)
        out.append(Case(cfg, mode, f'enc{i}', geom(op), 'b2s', big, small, role='fwd'))
        out.append(Case(cfg, mode, f'enc{i}', geom(op), 'wgrad', big, small))
        if i > 0:
            out.append(Case(cfg, mode, f'enc{i}', geom(op), 's2b', I(bfm), small, role='dgrad'))
    for i, op in enumerate(dec_ops):                     # ConvTranspose2d (-> InstanceNorm) -> act (unet.py:53-67)
        if i < 6:
            out.append(Case(cfg, mode, f'dec{i}', geom(op), 's2b', I(bfm), I(bfm), act='leakyrelu' if i == 0 else 'none', role='fwd'))
            out.append(Case(cfg, mode, f'dec{i}', geom(op), 'bwd_big', I(bfm), I(bfm)))
        else:                                            # the head: onto out_nc channels, sigmoid fused / softmax separate
            # (sigmoid: straight into the mask channels of the discriminator-input buffer; softmax: into a dense buffer first)
            gen = I(False, ld=3 + c['out_nc'], off=3) if c['final_act'] == 'sigmoid' else I(False)
            out.append(Case(cfg, mode, 'dec6', geom(op), 's2b', gen, I(bfm), act='sigmoid' if c['final_act'] == 'sigmoid' else 'none',
                            role='fwd'))
            out.append(Case(cfg, mode, 'dec6', geom(op), 'bwd_big', I(bfm, 1, pad8=bfm), I(bfm)))
    for nb, tag in ((B, 'N'), (2 * B, '2N')):            # D(fake) of the generator step at N, D(real | fake) at 2N (trainer.py:66,97-99)
        ops = de.ops(nb, S, S)
        last = len(ops) - 1
        for li, op in enumerate(ops):
            l = de.layers[li]
            name = f'd{li}/{tag}'
            if li == last:                               # 1-channel head: fp32 kernels on an fp32 copy of its input
                big, small = I(False), I(False)
            elif li == 0:
                big, small = I(bfm, 1, pad8=bfm), I(bfm)
            else:
                big, small = I(bfm), I(bfm)
            out.append(Case(cfg, mode, name, geom(op), 'b2s', big, small, bias=l.bias_key is not None, act=l.act, role='fwd'))
            if tag == '2N':
                out.append(Case(cfg, mode, name, geom(op), 'wgrad', big, small, bias=l.bias_key is not None))
            if li > 0 or tag == 'N':                     # data gradient; onto x | mask only in the generator step
                dbig = I(False) if li == 0 else I(bfm)   # (dL/d(x | mask) is fp32; the head writes the bf16 dy of the layer below)
                out.append(Case(cfg, mode, name, geom(op), 's2b', dbig, small, role='dgrad'))
    seen, uniq = set(), []
    for cs in out:
        k = (cs.geom, cs.op, cs.big.bf, cs.big.ldm, cs.big.pad8, cs.big.ld, cs.small.bf, cs.small.ldm, cs.bias, cs.act)
        if k not in seen:
            seen.add(k)
            uniq.append(cs)
    return uniq


def all_cases():
    return [cs for cfg, mode in MODES for cs in cases(cfg, mode)]


def live_plan():
    """{case key: [kernel symbols]} from the planner of the library that is loaded now."""
    return {cs.key: cs.symbols() for cs in all_cases()}


def committed_plan():
    with open(PLAN_FILE) as f:
        return json.load(f)
