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
