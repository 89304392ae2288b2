#!/usr/bin/env python3
"""Generates gemm4w_body_{nt,nn,tn}.inc: the hand-placed gfx950 main loop of the one-wave-per-SIMD bf16 GEMM (gemm4w.hip).

Workgroup = 256x256 output tile, 256 threads = 4 waves, ONE WAVE PER SIMD with the whole 512-entry register file; wave (wr, wc) owns the
128x128 quadrant (rows 128 wr.., columns 128 wc..): 64 accumulator tiles of 16x16 = a[0:255], tile (i, j) = a[4 (8 i + j) : +3], lane
(li = lane & 15, lq = lane >> 4) holding C[16 i + li][16 j + 4 lq + 0..3].  Per K tile of 64 a wave issues 128 v_mfma_f32_16x16x32_bf16
and reads ONE A half-tile + ONE B half-tile from LDS: 32 ds_read_b128 per 128 MFMAs -- 1.5 x fewer LDS bytes per FLOP than the 8-wave
kernel of gemm256.hip (24 per 64), the lever rule 28 of the guide ranks for a loop that is held down by power, and the structure of the
vendor kernel that sustains 1.64 PF/s on the box where gemm256 sustains 1.24 (profiles/r5_yardstick.txt).

Registers (hand-allocated; everything above v63 is left to the compiler):
  v[0:31]   A fragments A_i (i = 0..7: rows 16 i.. of the wave's quadrant) of the current k-step of 32
  v[32:63]  B fragments B_j (j = 0..7)
  a[0:255]  accumulators (physical-register asm outputs: the C++ epilogue reads them)
ONE fragment set: the MFMA order of a k-step is  [i = 0..7] x [j = 0..3]  then  [i = 0..7] x [j = 4..7], so A_i is free after MFMA 35 + 4 i,
B_j (j < 4) after MFMA 28 + j and B_j (j >= 4) after MFMA 60 + j - 4; each fragment of the NEXT k-step is re-read right behind the last use of
its register, 22..31 MFMAs (>= 350 cycles) before its first use.

LDS = 2 K-tile buffers x 4 half-tiles of 16 KiB, the images of gemm256.hip (KM: [128 rows][64 k], 16-byte slots XOR-swizzled; TR: [64 k][128
cols] for k-major operands, read with ds_read_b64_tr_b16), laid out so that the two buffers of a half-tile are 16 KiB apart: unit (slot, buffer)
at (2 slot + buffer) * 16384 with slot 0 = A rows 0..127, 1 = B cols 0..127, 2 = B cols 128..255, 3 = A rows 128..255 -- every LDS address of
the loop is a per-lane base register + an immediate.  Staging is LDS-DMA (buffer_load_dwordx4 ... lds), 16 pieces of 1 KiB per wave and K tile.

Synchronisation per K tile t (its second k-step): barrier ALPHA (gap 6: every wave has finished reading the A half-tiles of tile t) -> the 8 A
pieces of tile t + 2 go into the same buffer; barrier BETA (gap 31: the B half-tiles of tile t are read, tile t + 1 has landed: vmcnt(8) leaves
only the A pieces just issued in flight) -> the 8 B pieces of tile t + 2.  Every piece has >= 97 MFMAs (~1.5 k cycles) to land.
K tiles past the end of the contraction arrive as zeros (the descriptors' num_records are zeroed in the stream: s_live counts the tiles left).

The statement is entered with K tiles 0 and 1 of the output tile staged by the C++ wrapper (it issues them under the previous tile's epilogue)
and left with every accumulator complete and no memory operation outstanding.
"""
import os
import sys

FA, FB = 0, 32
N_HAND = 64
KINDS = {'nt': ('km', 'km'), 'nn': ('km', 'tr'), 'tn': ('tr', 'tr')}     # (A, B): km = k-contiguous rows, tr = k-major (transposed read)
SRD = {'A': 36, 'B': 40}          # pinned SGPR tuples s[36:39], s[40:43] (the stream zeroes word 2 = num_records at the end of the contraction)
ALPHA_GAP, BETA_GAP = 6, 31
OPTS = dict(kv.split('=') if '=' in kv else (kv, '1') for kv in filter(None, os.environ.get('G4W_OPTS', '').split(',')))
DROP = set(filter(None, os.environ.get('G4W_DROP', '').split(',')))      # timing-only builds (WRONG results): dsread dma barrier


def op(name):
    return f'%[{name}]'


def vregs(b, n):
    return {f'v{i}' for i in range(b, b + n)}


class I:
    def __init__(self, text, kind, reads=(), writes=(), tag=None):
        self.text, self.kind, self.reads, self.writes, self.tag = text, kind, set(reads), set(writes), tag


def acc(i, j):
    return 4 * (8 * i + j)


def mfma(i, j, zero_c):
    a = acc(i, j)
    c = '0' if zero_c else f'a[{a}:{a + 3}]'
    return I(f'v_mfma_f32_16x16x32_bf16 a[{a}:{a + 3}], v[{FB + 4 * j}:{FB + 4 * j + 3}], v[{FA + 4 * i}:{FA + 4 * i + 3}], {c}', 'mfma',
             reads=vregs(FA + 4 * i, 4) | vregs(FB + 4 * j, 4))


def kstep_order():
    return [(i, j) for i in range(8) for j in range(4)] + [(i, j) for i in range(8) for j in range(4, 8)]


class Gen:
    def __init__(self, layout):
        self.layout = layout
        self.kind = dict(zip('AB', KINDS[layout]))

    # ---- fragment reads
    def read_frag(self, opnd, idx, ks, b, tag):
        base = (FA if opnd == 'A' else FB) + 4 * idx
        if self.kind[opnd] == 'km':
            return [I(f'ds_read_b128 v[{base}:{base + 3}], {op(f"ar{opnd}k{ks}")} offset:{b * 16384 + idx * 2048}', 'ds', writes=vregs(base, 4), tag=tag)]
        off = b * 16384 + ks * 8192
        return [I(f'ds_read_b64_tr_b16 v[{base}:{base + 1}], {op(f"ar{opnd}t{idx}")} offset:{off}', 'ds', writes=vregs(base, 2), tag=tag),
                I(f'ds_read_b64_tr_b16 v[{base + 2}:{base + 3}], {op(f"ar{opnd}t{idx}")} offset:{off + 1024}', 'ds', writes=vregs(base + 2, 2), tag=tag)]

    def vops(self):
        out = []
        for o in 'AB':
            out += [f'ar{o}k0', f'ar{o}k1'] if self.kind[o] == 'km' else [f'ar{o}t{j}' for j in range(8)]
        return out + ['voffA', 'voffB']

    # ---- LDS-DMA pieces of one operand for K tile -> buffer b (slots: A0 = 0, B0 = 1, B1 = 2, A1 = 3)
    def dma_group(self, opnd, b):
        out = []
        slots = (0, 3) if opnd == 'A' else (1, 2)
        srd = SRD[opnd]
        for h, slot in enumerate(slots):
            for it in range(4):
                m0 = (2 * slot + b) * 16384 + it * 4096
                piece = [I(f's_add_u32 m0, {op("s_ldsw")}, {m0}', 'salu')]
                if h == 0 and it == 0:
                    piece.append(I('s_nop 0', 'salu'))
                    soff = op(f's_off{opnd}')
                else:
                    if it == 0:
                        piece.append(I(f's_add_u32 {op("s_t")}, {op(f"s_off{opnd}")}, {op(f"s_half{opnd}")}', 'salu'))
                    elif it == 1 and h == 0:
                        piece.append(I(f's_add_u32 {op("s_t")}, {op(f"s_off{opnd}")}, {op(f"s_it{opnd}")}', 'salu'))
                    else:
                        piece.append(I(f's_add_u32 {op("s_t")}, {op("s_t")}, {op(f"s_it{opnd}")}', 'salu'))
                    soff = op('s_t')
                piece.append(I(f'buffer_load_dwordx4 {op(f"voff{opnd}")}, s[{srd}:{srd + 3}], {soff} offen lds', 'dma'))
                out.append(piece)
        return out

    # ---- one k-step: 64 MFMAs + fillers
    def kstep(self, b, ks, zero_c, tile_tag):
        """tile in buffer b, k-step ks.  tag of a read = (tile_tag of the tile it reads, operand)"""
        bb = [mfma(i, j, zero_c) for (i, j) in kstep_order()]
        gaps = [[] for _ in range(64)]
        nb, nks, ntag = (b, 1, tile_tag) if ks == 0 else (1 - b, 0, tile_tag + 1)
        for jj, j in enumerate(range(4, 8)):                        # B_j (j >= 4) of THIS k-step (their registers were busy until MFMA 60.. of the previous one)
            gaps[1 + 4 * jj] += self.read_frag('B', j, ks, b, (tile_tag, 'B'))
        for j in range(4):
            gaps[33 + 4 * j] += self.read_frag('B', j, nks, nb, (ntag, 'B'))
        for i in range(8):
            gaps[35 + 4 * i] += self.read_frag('A', i, nks, nb, (ntag, 'A'))
        if ks == 1:
            # ALPHA: the A half-tiles of this tile are read by every wave -> restage them with tile + 2
            gaps[ALPHA_GAP] += [I('WAIT_TAG', 'waittag', tag=[(tile_tag, 'A')]),
                                I(f's_cmp_gt_i32 {op("s_live")}, 0', 'salu'),
                                I(f's_cselect_b32 s{SRD["A"] + 2}, s{SRD["A"] + 2}, 0', 'salu'),
                                I(f's_cselect_b32 s{SRD["B"] + 2}, s{SRD["B"] + 2}, 0', 'salu'),
                                I(f's_sub_u32 {op("s_live")}, {op("s_live")}, 1', 'salu'),
                                I('s_barrier', 'barrier')]
            a_gaps = [7, 10, 13, 16, 19, 22, 25, 28]
            for g, piece in zip(a_gaps, self.dma_group('A', b)):
                gaps[g] += piece
            gaps[a_gaps[-1]] += [I(f's_add_u32 {op("s_offA")}, {op("s_offA")}, {op("s_ktA")}', 'salu')]
            # BETA: the B half-tiles are read, the next tile has landed (all but the 8 A pieces just issued)
            gaps[BETA_GAP] += [I('WAIT_TAG', 'waittag', tag=[(tile_tag, 'B'), (tile_tag, 'A')], ),
                               I('s_waitcnt vmcnt(8)', 'vmwait'),
                               I('s_barrier', 'barrier')]
            b_gaps = [34, 38, 42, 46, 50, 54, 58, 62]
            for g, piece in zip(b_gaps, self.dma_group('B', b)):
                gaps[g] += piece
            gaps[b_gaps[-1]] += [I(f's_add_u32 {op("s_offB")}, {op("s_offB")}, {op("s_ktB")}', 'salu')]
        return bb, gaps

    def pair(self, first, tag0):
        """K tiles (tag0 -> buffer 0, tag0 + 1 -> buffer 1)"""
        out = []
        for b in range(2):
            for ks in range(2):
                out.append(self.kstep(b, ks, first and b == 0 and ks == 0, tag0 + b))
        return out


class Hazards:
    """linear walk: counted lgkmcnt waits for LDS reads before their consumers (merged: a wait also covers every older-than-OLD read)"""
    OLD = 12      # a read issued >= OLD MFMAs ago has certainly returned: waiting for it costs nothing

    def __init__(self):
        self.out = []
        self.pending = []      # (writes, tag, mfma index at issue)
        self.n_mfma = 0

    def _wait_for(self, k):
        """wait until pending[0..k] have returned; extend k over reads that are old anyway"""
        while k + 1 < len(self.pending) and self.n_mfma - self.pending[k + 1][2] >= self.OLD:
            k += 1
        after = len(self.pending) - 1 - k
        self.out.append(I(f's_waitcnt lgkmcnt({min(after, 15)})', 'wait'))
        self.pending = self.pending[k + 1:] if after <= 15 else []

    def emit(self, ins):
        cls = {'ds': 'dsread', 'dma': 'dma', 'barrier': 'barrier', 'vmwait': 'vmwait'}.get(ins.kind)
        if cls in DROP and self.in_loop:
            return
        if ins.kind == 'waittag':
            need = None
            for k, (wr, tag, _) in enumerate(self.pending):
                if tag in ins.tag:
                    need = k
            if need is not None:
                self._wait_for(need)
            return
        used = ins.reads | ins.writes
        need = None
        for k, (wr, tag, _) in enumerate(self.pending):
            if wr & used:
                need = k
        if need is not None:
            self._wait_for(need)
        self.out.append(ins)
        if ins.kind == 'ds':
            self.pending.append((ins.writes, ins.tag, self.n_mfma))
        if ins.kind == 'mfma':
            self.n_mfma += 1

    in_loop = False

    def drain(self, text):
        self.out.append(I(text, 'wait'))
        self.pending = []


def generate(layout):
    G = Gen(layout)
    H = Hazards()
    E = H.emit
    # ---- entry: K tiles 0 and 1 were issued by the wrapper; every wave's pieces landed -> barrier -> fragments of (tile 0, k-step 0)
    H.drain('s_waitcnt vmcnt(0) lgkmcnt(0)')
    E(I('s_barrier', 'barrier'))
    for j in range(4):
        for ins in G.read_frag('B', j, 0, 0, (0, 'B')):
            E(ins)
    for i in range(8):
        for ins in G.read_frag('A', i, 0, 0, (0, 'A')):
            E(ins)

    def emit_pair(first, tag0):
        for bb, gaps in G.pair(first, tag0):
            for g, m in enumerate(bb):
                E(m)
                for ins in gaps[g]:
                    E(ins)

    H.in_loop = True
    emit_pair(True, 0)
    E(I(f's_sub_u32 {op("s_cnt")}, {op("s_cnt")}, 1', 'salu'))
    E(I(f's_cmp_eq_u32 {op("s_cnt")}, 0', 'salu'))
    E(I('s_cbranch_scc1 EXIT%=', 'branch'))
    # the loop body starts from the same pending-read state as the code behind the first pair (tags are relative: shift by 2 per pair)
    state = [(w, (t[0] - 2, t[1]), n - H.n_mfma) for (w, t, n) in H.pending]
    H.out.append(I('LOOP%=:', 'label'))
    H.pending = [(w, t, n + H.n_mfma) for (w, t, n) in state]
    emit_pair(False, 0)
    end_state = [(w, (t[0] - 2, t[1]), n - H.n_mfma) for (w, t, n) in H.pending]
    assert [(sorted(w), t) for (w, t, n) in end_state] == [(sorted(w), t) for (w, t, n) in state], 'loop-carried LDS state differs'
    E(I(f's_sub_u32 {op("s_cnt")}, {op("s_cnt")}, 1', 'salu'))
    E(I(f's_cmp_eq_u32 {op("s_cnt")}, 0', 'salu'))
    E(I('s_cbranch_scc0 LOOP%=', 'branch'))
    H.in_loop = False
    H.out.append(I('EXIT%=:', 'label'))
    H.drain('s_waitcnt vmcnt(0) lgkmcnt(0)')
    H.out.append(I('s_barrier', 'barrier'))          # every wave has left the LDS: the wrapper may stage the next output tile
    H.out.append(I('s_nop 7\n\ts_nop 7', 'nop'))
    return H.out, G


def render(stream, G):
    lines = []
    for ins in stream:
        lines += ins.text.split('\n\t')
    body = '\n'.join(f'    "{ln}\\n\\t"' for ln in lines)
    accs = ', '.join(f'"={{a[{acc(i, j)}:{acc(i, j) + 3}]}}"(c[{i}][{j}])' for i in range(8) for j in range(8))
    srw = ['s_offA', 's_offB', 's_live', 's_cnt']
    outs = accs + ',\n      ' + f'"+{{s[{SRD["A"]}:{SRD["A"] + 3}]}}"(srdA), "+{{s[{SRD["B"]}:{SRD["B"] + 3}]}}"(srdB),\n      ' + \
        ', '.join(f'[{n}] "+&s"({n})' for n in srw) + ', [s_t] "=&s"(s_t)'
    sin = ['s_ldsw', 's_itA', 's_halfA', 's_ktA', 's_itB', 's_halfB', 's_ktB']
    ins_ = ', '.join(f'[{n}] "v"({n})' for n in G.vops()) + ',\n      ' + ', '.join(f'[{n}] "s"({n})' for n in sin)
    clob = ', '.join(f'"v{i}"' for i in range(N_HAND)) + ', "vcc", "scc", "memory"'
    return (f'// GENERATED by gen_gemm4w.py ({G.layout}) -- do not edit; see that file for the register map and the schedule\n'
            'asm volatile(\n' + body + '\n    : ' + outs + '\n    : ' + ins_ + '\n    : ' + clob + ');\n')


if __name__ == '__main__':
    here = os.path.dirname(os.path.abspath(__file__))
    for layout in KINDS:
        stream, G = generate(layout)
        with open(os.path.join(here, f'gemm4w_body_{layout}.inc'), 'w') as f:
            f.write(render(stream, G))
        if '-v' in sys.argv:
            kinds = {}
            for ins in stream:
                kinds[ins.kind] = kinds.get(ins.kind, 0) + 1
            print(layout, kinds)
