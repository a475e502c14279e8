"""Kernel metadata of the gfx950 code objects embedded in libpatchgan_hip.so (no GPU, no ROCm tool needed): the .hip_fatbin
section holds one clang offload bundle per translation unit; each bundle entry for the device target is an ELF whose
NT_AMDGPU_METADATA note is a msgpack map with one record per kernel (.name, .vgpr_count, .private_segment_fixed_size, ...)."""
import struct

import msgpack

MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def _sections(elf):
    shoff, = struct.unpack_from('<Q', elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from('<HHH', elf, 0x3A)
    secs = []
    for i in range(shnum):
        name, typ, flags, addr, off, size = struct.unpack_from('<IIQQQQ', elf, shoff + i * shentsize)
        secs.append((name, typ, off, size))
    stroff = secs[shstrndx][2]
    out = {}
    for name, typ, off, size in secs:
        end = elf.index(b'\0', stroff + name)
        out.setdefault(elf[stroff + name:end].decode(), []).append((typ, off, size))
    return out


def _notes(elf):
    for secs in _sections(elf).values():
        for typ, off, size in secs:
            if typ != 7:        # SHT_NOTE
                continue
            p = off
            while p + 12 <= off + size:
                namesz, descsz, ntype = struct.unpack_from('<III', elf, p)
                p += 12
                name = elf[p:p + namesz].rstrip(b'\0')
                p += (namesz + 3) & ~3
                desc = elf[p:p + descsz]
                p += (descsz + 3) & ~3
                yield name, ntype, desc


def kernels(so_path, arch='gfx950'):
    """{demangled-ish kernel symbol: metadata dict} over every device code object for `arch` in the shared library."""
    data = open(so_path, 'rb').read()
    out = {}
    for typ, off, size in _sections(data).get('.hip_fatbin', []):
        blob = data[off:off + size]
        p = blob.find(MAGIC)
        while p >= 0:
            n, = struct.unpack_from('<Q', blob, p + len(MAGIC))
            q = p + len(MAGIC) + 8
            for _ in range(n):
                eoff, esize, tsize = struct.unpack_from('<QQQ', blob, q)
                triple = blob[q + 24:q + 24 + tsize].decode()
                q += 24 + tsize
                if arch in triple and esize:
                    co = blob[p + eoff:p + eoff + esize]
                    for name, ntype, desc in _notes(co):
                        if name == b'AMDGPU' and ntype == 32:
                            for k in msgpack.unpackb(desc, raw=False).get('amdhsa.kernels', []):
                                out[k['.name']] = k
            p = blob.find(MAGIC, p + 1)
    return out


def code_objects(so_path, arch='gfx950'):
    """The device ELF images for `arch` embedded in the shared library (one per translation unit)."""
    data = open(so_path, 'rb').read()
    out = []
    for typ, off, size in _sections(data).get('.hip_fatbin', []):
        blob = data[off:off + size]
        p = blob.find(MAGIC)
        while p >= 0:
            n, = struct.unpack_from('<Q', blob, p + len(MAGIC))
            q = p + len(MAGIC) + 8
            for _ in range(n):
                eoff, esize, tsize = struct.unpack_from('<QQQ', blob, q)
                triple = blob[q + 24:q + 24 + tsize].decode()
                q += 24 + tsize
                if arch in triple and esize:
                    out.append(blob[p + eoff:p + eoff + esize])
            p = blob.find(MAGIC, p + 1)
    return out


_REG = None


def _vregs(operand):
    import re
    global _REG
    if _REG is None:
        _REG = re.compile(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b')
    out = []
    for m in _REG.finditer(operand):
        if m.group(3) is not None:
            out.append(int(m.group(3)))
        else:
            out.extend(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def packed_f32_reads_of_fresh_lds_pairs(disassembly_lines):
    """The operand form behind round 3's open determinism issue (EXPERIMENTS.md, "k_s2b_ca1_s1: cause"): a packed-f32 VALU instruction
    (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) that takes the LOW half of a source from the ODD register of a pair (op_sel bit 1)
    while that register's last writer is an LDS read.  Yields (function, instruction).  Linear walk per function (no control flow):
    a conservative reading of "last writer"."""
    import re
    opsel = re.compile(r'op_sel:\[([01,]+)\]')
    fn, fresh = None, set()
    for line in disassembly_lines:
        if line.startswith('0000') and line.rstrip().endswith('>:'):
            fn, fresh = line.split('<')[1].split('>')[0], set()
            continue
        s = line.split('//')[0].strip()
        if not s or fn is None:
            continue
        parts = s.split(None, 1)
        op, args = parts[0], (parts[1] if len(parts) > 1 else '')
        ops = [a.strip() for a in args.split(',')]
        if op.startswith('ds_read') or op.startswith('ds_load'):
            fresh.update(_vregs(ops[0]))
            continue
        if op.startswith('v_pk_') and op.endswith('_f32'):
            m = opsel.search(s)
            if m:
                sel = [int(x) for x in m.group(1).split(',')]
                srcs = [o.split(' ')[0] for o in ops[1:1 + len(sel)]]
                for o, sl in zip(srcs, sel):
                    r = _vregs(o)
                    if sl == 1 and len(r) == 2 and r[1] in fresh:
                        yield fn, s
                        break
        if op.startswith(('v_', 'global_load', 'flat_load', 'scratch_load')) or (op.startswith('buffer_load') and ' lds' not in args):
            if ops:
                fresh.difference_update(_vregs(ops[0]))
