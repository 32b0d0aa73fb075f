"""A progressive JPEG writer for the decoder tests: quantised coefficients in, a file with ANY legal scan script out.  Pillow's
writer (libjpeg-turbo) only ever produces libjpeg's default script; files from other encoders (mozjpeg: what most image hosts
serve) order and split their scans differently -- DC scans for one, two or all components, spectral bands cut anywhere,
successive approximation from any bit down -- and the decoder has to be held against Pillow on those too.  The entropy coding
follows T.81 Annex G as libjpeg's jcphuff.c does (one end-of-band symbol per block, no end-of-band runs; the standard Huffman
tables of Annex K, taken from a file Pillow writes, carry every symbol that needs)."""
from __future__ import annotations

import io
import struct

import numpy as np
from PIL import Image


def _standard_tables():
    """{(class, id): {symbol: (code, length)}} from the DHT segments of a baseline file Pillow writes without optimisation."""
    b = io.BytesIO()
    Image.fromarray(np.zeros((16, 16, 3), np.uint8)).save(b, "JPEG", quality=75)
    data = b.getvalue()
    tables, raw = {}, {}
    pos = 2
    while pos < len(data) and data[pos] == 0xFF and data[pos + 1] != 0xDA:
        marker, n = data[pos + 1], struct.unpack(">H", data[pos + 2:pos + 4])[0]
        if marker == 0xC4:
            at, end = pos + 4, pos + 2 + n
            while at < end:
                tc_th = data[at]
                counts = list(data[at + 1:at + 17])
                syms = list(data[at + 17:at + 17 + sum(counts)])
                raw[(tc_th >> 4, tc_th & 15)] = bytes(data[at:at + 17 + sum(counts)])
                code, k, table = 0, 0, {}
                for length in range(1, 17):
                    for _ in range(counts[length - 1]):
                        table[syms[k]] = (code, length)
                        code += 1
                        k += 1
                    code <<= 1
                tables[(tc_th >> 4, tc_th & 15)] = table
                at += 17 + sum(counts)
        pos += 2 + n
    return tables, raw


_TABLES, _RAW = _standard_tables()


class _Bits:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, value: int, length: int):
        if length == 0:
            return
        self.acc = (self.acc << length) | (value & ((1 << length) - 1))
        self.n += length
        while self.n >= 8:
            byte = (self.acc >> (self.n - 8)) & 255
            self.out.append(byte)
            if byte == 0xFF:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def finish(self) -> bytes:
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)          # pad with ones
        return bytes(self.out)


def _category(v: int) -> int:
    return int(abs(v)).bit_length()


def _magnitude_bits(v: int, size: int) -> int:
    return v if v >= 0 else v + (1 << size) - 1


def encode(width: int, height: int, comps, coefs, script, qtables=None) -> bytes:
    """comps: [(id, hs, vs)]; coefs[c]: int array [padded block rows][padded block columns][64] in zigzag order (padded to whole
    MCUs); script: [(component indices, ss, se, ah, al)] -- any order T.81 allows."""
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    mcus_x, mcus_y = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    out = bytearray(b"\xff\xd8" + b"\xff\xe0" + struct.pack(">H5sBBBHHBB", 16, b"JFIF\0", 1, 1, 0, 1, 1, 0, 0))
    qtables = qtables or [np.ones(64, np.uint8), np.ones(64, np.uint8)]
    for k, q in enumerate(qtables):
        out += b"\xff\xdb" + struct.pack(">HB", 67, k) + bytes(int(v) for v in q)
    out += b"\xff\xc2" + struct.pack(">HBHHB", 8 + 3 * len(comps), 8, height, width, len(comps))
    for k, (cid, hs, vs) in enumerate(comps):
        out += bytes([cid, (hs << 4) | vs, 0 if k == 0 else 1])
    for key in ((0, 0), (1, 0)):
        out += b"\xff\xc4" + struct.pack(">H", 2 + len(_RAW[key])) + _RAW[key]
    dc, ac = _TABLES[(0, 0)], _TABLES[(1, 0)]
    for (members, ss, se, ah, al) in script:
        out += b"\xff\xda" + struct.pack(">HB", 6 + 2 * len(members), len(members))
        for c in members:
            out += bytes([comps[c][0], 0x00])
        out += bytes([ss, se, (ah << 4) | al])
        bits = _Bits()
        if ss == 0:
            pred = [0] * len(comps)
            if len(members) > 1:
                units = [(my, mx) for my in range(mcus_y) for mx in range(mcus_x)]
            else:
                c = members[0]
                cw, ch = -(-width * comps[c][1] // hmax), -(-height * comps[c][2] // vmax)
                units = [(by, bx) for by in range(-(-ch // 8)) for bx in range(-(-cw // 8))]
            for (uy, ux) in units:
                for c in members:
                    hs, vs = (comps[c][1], comps[c][2]) if len(members) > 1 else (1, 1)
                    for by in range(vs):
                        for bx in range(hs):
                            v = int(coefs[c][uy * vs + by, ux * hs + bx, 0])
                            if ah == 0:
                                t = v >> al                                  # arithmetic shift, as jcphuff does for DC
                                diff = t - pred[c]
                                pred[c] = t
                                size = _category(diff)
                                bits.put(*dc[size])
                                bits.put(_magnitude_bits(diff, size), size)
                            else:
                                bits.put((v >> al) & 1, 1)
        else:
            c = members[0]
            cw, ch = -(-width * comps[c][1] // hmax), -(-height * comps[c][2] // vmax)
            for by in range(-(-ch // 8)):
                for bx in range(-(-cw // 8)):
                    blk = coefs[c][by, bx]
                    if ah == 0:
                        run = 0
                        for k in range(ss, se + 1):
                            v = int(blk[k])
                            t = (abs(v) >> al) * (1 if v >= 0 else -1)
                            if t == 0:
                                run += 1
                                continue
                            while run > 15:
                                bits.put(*ac[0xF0])
                                run -= 16
                            size = _category(t)
                            bits.put(*ac[(run << 4) | size])
                            bits.put(_magnitude_bits(t, size), size)
                            run = 0
                        if run > 0:
                            bits.put(*ac[0x00])
                    else:
                        absval = [abs(int(blk[k])) >> al for k in range(64)]
                        eob = max([k for k in range(ss, se + 1) if absval[k] == 1], default=-1)
                        run, pending = 0, []
                        for k in range(ss, se + 1):
                            t = absval[k]
                            if t == 0:
                                run += 1
                                continue
                            while run > 15 and k <= eob:
                                bits.put(*ac[0xF0])
                                for b in pending:
                                    bits.put(b, 1)
                                pending = []
                                run -= 16
                            if t > 1:                                        # already nonzero: one more bit of it, sent behind the next symbol
                                pending.append(t & 1)
                                continue
                            bits.put(*ac[(run << 4) | 1])
                            bits.put(1 if int(blk[k]) >= 0 else 0, 1)
                            for b in pending:
                                bits.put(b, 1)
                            pending = []
                            run = 0
                        if run > 0 or pending:
                            bits.put(*ac[0x00])
                            for b in pending:
                                bits.put(b, 1)
        out += bits.finish()
    return bytes(out) + b"\xff\xd9"


def random_script(rng, ncomp: int):
    """A random legal progression: DC scans for all, some or single components from a random bit down; for every component its
    AC coefficients cut into 1..4 bands, each from a random bit down; scans shuffled as far as the rules allow (a component's
    first DC scan before its AC scans, a band's refinements in order)."""
    chains = []
    al0 = int(rng.integers(0, 3))
    groups = [list(range(ncomp))] if ncomp == 1 or rng.integers(0, 2) else ([[0], list(range(1, ncomp))] if rng.integers(0, 2) else [[c] for c in range(ncomp)])
    dc_first = [(g, 0, 0, 0, al0) for g in groups]
    for bit in range(al0, 0, -1):
        g2 = [list(range(ncomp))] if rng.integers(0, 2) else [[c] for c in range(ncomp)]
        chains.append([(g, 0, 0, bit, bit - 1) for g in g2])
    # refinements of DC must follow each other in order: one chain
    dc_chain = [s for level in chains for s in level]
    chains = [dc_chain] if dc_chain else []
    for c in range(ncomp):
        cuts = sorted(set(rng.integers(2, 64, int(rng.integers(0, 4))).tolist()))
        edges = [1] + cuts + [64]
        for lo, hi in zip(edges[:-1], edges[1:]):
            al = int(rng.integers(0, 3))
            chain = [([c], lo, hi - 1, 0, al)] + [([c], lo, hi - 1, bit, bit - 1) for bit in range(al, 0, -1)]
            chains.append(chain)
    script = list(dc_first)
    rng.shuffle(script)
    heads = [0] * len(chains)
    while any(h < len(ch) for h, ch in zip(heads, chains)):
        k = int(rng.choice([i for i, (h, ch) in enumerate(zip(heads, chains)) if h < len(ch)]))
        script.append(chains[k][heads[k]])
        heads[k] += 1
    return script


def random_file(rng, width: int, height: int, sampling: str = "444", gray: bool = False):
    """(file bytes, script) for random smooth-ish coefficients."""
    comps = [(1, 1, 1)] if gray else [(1, {"444": 1, "422": 2, "420": 2, "440": 1}[sampling], {"444": 1, "422": 1, "420": 2, "440": 2}[sampling]), (2, 1, 1), (3, 1, 1)]
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    mcus_x, mcus_y = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    coefs = []
    for (_, hs, vs) in comps:
        shape = (mcus_y * vs, mcus_x * hs, 64)
        a = np.zeros(shape, np.int32)
        a[:, :, 0] = rng.integers(-60, 61, shape[:2]) * 8
        decay = np.maximum(1, (40 / (1 + np.arange(64))).astype(np.int32))
        a[:, :, 1:] = (rng.integers(-1, 2, (shape[0], shape[1], 63)) * rng.integers(0, decay[1:] + 1, (shape[0], shape[1], 63))) * (rng.random((shape[0], shape[1], 63)) < 0.35)
        coefs.append(a)
    q = [np.full(64, 2, np.uint8), np.full(64, 3, np.uint8)]
    script = random_script(rng, len(comps))
    return encode(width, height, comps, coefs, script, q), script
