"""Per-kernel parity tests of libcruller_hip.so on a real MI355X (run with `-m gpu`).

Every test calls the HIP kernel through the C-ABI (pixparse_amd.ops -> ctypes) and compares with a
plain fp32 PyTorch restatement of the same op (computed with torch on the same device, or by the CPU
oracle).  bf16 results are compared with tolerances that are stated next to each check.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

BF16, F16, F32 = torch.bfloat16, torch.float16, torch.float32


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    from pixparse_amd import hip
    hip.load()
    return torch.device('cuda:0')


def rnd(shape, dev, scale=1.0, seed=0, dtype=F32):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev).to(dtype)


def gelu_grad_ref(h):
    """erf-GELU derivative Phi(h) + h phi(h), evaluated in fp64 (what the GELU epilogue saves as fp16 and the dgrad epilogue multiplies by)"""
    h64 = h.double()
    return (0.5 * (1.0 + torch.erf(h64 / math.sqrt(2.0))) + h64 * torch.exp(-0.5 * h64 * h64) / math.sqrt(2.0 * math.pi)).float()


def close(a, b, rtol, atol, what=''):
    a, b = a.float(), b.float()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bad.any(), f'{what}: {int(bad.sum())}/{bad.numel()} off, max err {float(err.max()):.4g} (ref max {float(b.abs().max()):.4g})'


# ------------------------------------------------------------------------------------------- attention with a prescaled q
@pytest.mark.parametrize('B,H,Nq,Nk,causal', [(1, 2, 200, 333, False), (2, 1, 130, 130, True), (1, 2, 64, 1500, False), (1, 1, 300, 45, False)])
def test_attention_q_prescaled(dev, B, H, Nq, Nk, causal):
    """q_prescaled = 1: q carries softmax_scale * log2(e); the forward is the seeded / lazy-maximum kernel (first tile and any tile whose
    row sums exceed 2^30 re-centre, every other tile runs without a row maximum).  Forward and backward against fp32 torch on the SAME
    (prescaled) operands: out, lse, dq (gradient of the UNscaled projection output), dk, dv; and against the unscaled kernels fed the
    exact same mathematical problem.  A second data set makes later tiles overflow the reference of the first (scores growing along the
    keys) so that the re-centring path runs in the middle of a row."""
    from pixparse_amd import ops
    D, scale = H * 64, 0.125
    c = scale * ops.LOG2E
    for growing in (False, True):
        g = torch.Generator(device=dev).manual_seed(Nq + Nk + int(growing))
        qraw = torch.randn(B, Nq, D, generator=g, device=dev)
        k = torch.randn(B, Nk, D, generator=g, device=dev)
        if growing:         # later keys score ever higher: up to ~2^40 above the first tile's maximum along a row
            k = k * torch.linspace(0.2, 6.0, Nk, device=dev).view(1, Nk, 1)
            qraw = qraw.abs() * 1.5
            k = k.abs()
        k = k.to(BF16)
        v, do = (torch.randn(B, t, D, generator=g, device=dev).to(BF16) for t in (Nk, Nq))
        qpre = (qraw * c).to(BF16)                                # what the projection GEMM's colscale epilogue writes
        o = torch.empty_like(qpre)
        lse = torch.empty(B, H, Nq, device=dev)
        ops.attn_fwd(qpre, k, v, o, lse, H, scale, causal, q_prescaled=True)
        dq, dk, dv = torch.full_like(qpre, float('nan')), torch.full_like(k, float('nan')), torch.full_like(v, float('nan'))
        delta = torch.empty(2, B, H, Nq, device=dev)
        ops.attn_bwd(qpre, k, v, o, do, lse, delta, dq, dk, dv, H, scale, causal, q_prescaled=True)
        hd = lambda t: t.float().reshape(B, -1, H, 64).transpose(1, 2)
        qproj = (hd(qpre) / c).requires_grad_(True)              # the unscaled projection output whose scaled copy the kernels saw
        K, V = hd(k).requires_grad_(True), hd(v).requires_grad_(True)
        z = (qproj * c) @ K.transpose(-1, -2) * math.log(2.0)     # natural-log logits = ln 2 * base-2 logits
        if causal:
            mask = torch.ones(Nq, Nk, dtype=torch.bool, device=dev).tril(diagonal=Nk - Nq)
            z = z.masked_fill(~mask, float('-inf'))
        out = torch.softmax(z, -1) @ V
        out.backward(hd(do))
        back = lambda t: t.transpose(1, 2).reshape(B, -1, D)
        rel = lambda a, b: float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))
        assert rel(o, back(out.detach())) < 1e-2, (growing, rel(o, back(out.detach())))
        assert float((lse - torch.logsumexp(z, -1)).abs().max()) < 3e-3 * max(1.0, float(z.abs().max()) / 50)
        assert rel(dq, back(qproj.grad)) < 2e-2 and rel(dk, back(K.grad)) < 2e-2 and rel(dv, back(V.grad)) < 2e-2, \
            (growing, rel(dq, back(qproj.grad)), rel(dk, back(K.grad)), rel(dv, back(V.grad)))
        if not growing:      # the unscaled kernels on the same problem (q = qpre / c is not bf16-exact: compare at bf16 tolerance)
            o2 = torch.empty_like(qpre)
            lse2 = torch.empty_like(lse)
            ops.attn_fwd((qpre.float() / c).to(BF16), k, v, o2, lse2, H, scale, causal)
            assert rel(o, o2) < 2e-2


@pytest.mark.parametrize('B,H,Nq,Nk', [(1, 2, 200, 333), (1, 1, 64, 128), (2, 8, 700, 1000), (1, 2, 1023, 1300), (1, 1, 45, 6189), (1, 3, 513, 192), (1, 2, 256, 2049)])
def test_attention_fwd_one_wave_per_simd(dev, B, H, Nq, Nk):
    """the hand-placed forward stream (crl_attn_fwd_set_mode 0: 256 queries per workgroup, fixed first-tile reference, row sums on the matrix
    pipe, masked last key tile) against fp32 torch and against the compiler-scheduled kernel (mode 1) on ragged Nq / Nk; mode 2 (every block
    re-run by its moving-maximum fallback) must reproduce mode 1 bit for bit; twice the same bits; the one-workgroup-per-CU form (mode 3) the same bits"""
    from pixparse_amd import hip, ops
    D = H * 64
    c = 0.125 * ops.LOG2E
    g = torch.Generator(device=dev).manual_seed(Nq * 7 + Nk)
    qpre = (torch.randn(B, Nq, D, generator=g, device=dev) * c * 2.0).to(BF16)
    k = torch.randn(B, Nk, D, generator=g, device=dev).to(BF16)
    v = torch.randn(B, Nk, D, generator=g, device=dev).to(BF16)
    res = {}
    try:
        for mode in (0, 1, 2, 0, 3):
            hip.call('crl_attn_fwd_set_mode', mode)
            o = torch.full_like(qpre, float('nan'))
            lse = torch.full((B, H, Nq), float('nan'), device=dev)
            ops.attn_fwd(qpre, k, v, o, lse, H, 0.125, False, q_prescaled=True)
            if mode in res:
                assert torch.equal(res[mode][0], o) and torch.equal(res[mode][1], lse), 'not reproducible'
            res[mode] = (o, lse)
    finally:
        hip.call('crl_attn_fwd_set_mode', 0)
    hd = lambda t: t.float().reshape(B, -1, H, 64).transpose(1, 2)
    z = hd(qpre) @ hd(k).transpose(-1, -2) * math.log(2.0)
    ref = (torch.softmax(z, -1) @ hd(v)).transpose(1, 2).reshape(B, Nq, D)
    lref = torch.logsumexp(z, -1)
    rel = lambda a, b: float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))
    assert torch.equal(res[2][0], res[1][0]) and torch.equal(res[2][1], res[1][1]), 'fallback differs from the compiler-scheduled kernel'
    assert torch.equal(res[3][0], res[0][0]) and torch.equal(res[3][1], res[0][1]), 'the 512-register form differs from the 256-register form'
    assert rel(res[0][0], ref) < 1e-2 and rel(res[1][0], ref) < 1e-2, (rel(res[0][0], ref), rel(res[1][0], ref))
    # lse: the stream sums the bf16-ROUNDED probabilities (the operand of P.V, on the matrix pipe): a row dominated by one key carries that key's
    # rounding, half a bf16 ulp = 2^-8 = 3.9e-3 relative in l at most; the normalised weights of the output sum to one exactly
    assert float((res[0][1] - lref).abs().max()) < 5e-3, float((res[0][1] - lref).abs().max())
    assert float((res[1][1] - lref).abs().max()) < 3e-3, float((res[1][1] - lref).abs().max())
    close(res[0][0], ref, 2e-2, 2e-2, 'stream out')


def test_attention_stream_lse_on_peaked_rows(dev):
    """the forward stream's lse is the log of the sum of the bf16-ROUNDED probabilities.  Worst case = rows that one key dominates (the
    dominant probability is rounded by up to half a bf16 ulp and nothing averages it out): the probabilities the backward rebuilds from that lse,
    P = exp(z - lse), must still sum to one within 2^-8 (+ slack), on every row; rows with flat scores sit within 1e-3; and the compiler-scheduled
    kernel (mode 1: fp32 row sums) stays within fp32 rounding on the same data -- the documented bound of include/crl.h"""
    from pixparse_amd import hip, ops
    B, H, Nq, Nk = 1, 2, 512, 1024
    D, c = H * 64, 0.125 * ops.LOG2E
    g = torch.Generator(device=dev).manual_seed(17)
    k = torch.randn(B, Nk, D, generator=g, device=dev)
    q = torch.randn(B, Nq, D, generator=g, device=dev)
    # rows 0..255: query i = 9 x (key i): that key's score exceeds the others' by ~ 9 |k|^2 / 8 ~ 70 nats -> one probability ~ 1, the rest ~ 0
    # rows 256..383: two keys share the row (two rounded probabilities near 0.5);  rows 384..: random (flat) scores
    q[:, :256] = 9.0 * k[:, :256]
    q[:, 256:384] = 2.0 * (k[:, 256:384] + k[:, 600:728])
    qpre, kb = (q * c).to(BF16), k.to(BF16)
    v = torch.randn(B, Nk, D, generator=g, device=dev).to(BF16)
    hd = lambda t: t.float().reshape(B, -1, H, 64).transpose(1, 2)
    z = hd(qpre) @ hd(kb).transpose(-1, -2) * math.log(2.0)
    worst = {}
    for mode in (0, 1):
        hip.call('crl_attn_fwd_set_mode', mode)
        try:
            o, lse = torch.empty_like(qpre), torch.empty(B, H, Nq, device=dev)
            ops.attn_fwd(qpre, kb, v, o, lse, H, 0.125, False, q_prescaled=True)
        finally:
            hip.call('crl_attn_fwd_set_mode', 0)
        assert torch.isfinite(lse).all() and torch.isfinite(o.float()).all()
        dev_sum = (torch.exp(z - lse[..., None]).sum(-1) - 1.0).abs()          # |sum_k P_bwd - 1| per row
        worst[mode] = (float(dev_sum[:, :, :384].max()), float(dev_sum[:, :, 384:].max()))
    assert worst[0][0] < 2.0 ** -8 + 5e-4, worst        # peaked rows: half a bf16 ulp of the dominant probability
    assert worst[0][1] < 1e-3, worst                    # flat rows: the roundings average out
    assert worst[1][0] < 1e-3 and worst[1][1] < 1e-3, worst      # fp32 row sums: fp32 rounding of scores up to ~100 nats


def test_attention_fwd_persistent_launch_pulls_the_same_blocks(dev):
    """crl_attn_fwd_set_persistent: with more query blocks than workgroup slots (all but 32 CUs reserved: 64 slots) the stream is launched persistently and
    pulls its blocks from the per-XCD ticket lists -- bit-identical to one workgroup per block, also with the static walk and in the one-per-CU form;
    forty launches through the slot ring leave the counters zeroed"""
    from pixparse_amd import hip, ops
    B, H, Nq, Nk = 4, 8, 2100, 600           # 9 query blocks x 32 heads = 288 items for 64 slots (32 CUs in the one-per-CU form)
    D = H * 64
    g = torch.Generator(device=dev).manual_seed(11)
    qpre = (torch.randn(B, Nq, D, generator=g, device=dev) * 0.125 * ops.LOG2E * 2.0).to(BF16)
    k = torch.randn(B, Nk, D, generator=g, device=dev).to(BF16)
    v = torch.randn(B, Nk, D, generator=g, device=dev).to(BF16)

    def fwd(persist, mode=0, dynamic=True):
        hip.call('crl_attn_fwd_set_persistent', persist)
        hip.call('crl_attn_fwd_set_mode', mode)
        ops.gemm_set_schedule(dynamic)
        o = torch.full_like(qpre, float('nan'))
        lse = torch.full((B, H, Nq), float('nan'), device=dev)
        ops.attn_fwd(qpre, k, v, o, lse, H, 0.125, False, q_prescaled=True)
        return o, lse
    try:
        ops.gemm_set_reserved_cus(224)
        want = fwd(0)
        for rep in range(40):
            got = fwd(1)
            if rep in (0, 39):
                assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), f'persistent launch differs (rep {rep})'
        for got in (fwd(1, dynamic=False), fwd(1, mode=3), fwd(0, mode=3)):
            assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    finally:
        ops.gemm_set_reserved_cus(0)
        ops.gemm_set_schedule(True)
        hip.call('crl_attn_fwd_set_persistent', 1)
        hip.call('crl_attn_fwd_set_mode', 0)


def test_attention_fwd_stream_overflow_falls_back(dev):
    """rows whose later scores exceed the first key tile's maximum by more than 2^127 (and rows that stay far below it) -- the stream's fixed
    reference overflows, the workgroup re-runs the block with the moving maximum: finite and right; moderate growth (2^60) stays in line"""
    from pixparse_amd import hip, ops
    B, H, Nq, Nk = 1, 2, 300, 640
    D = H * 64
    for amp, must_equal_mode1 in ((40.0, False), (400.0, True)):
        g = torch.Generator(device=dev).manual_seed(int(amp))
        q = torch.randn(B, Nq, D, generator=g, device=dev).abs()
        k = torch.randn(B, Nk, D, generator=g, device=dev).abs() * torch.linspace(0.05, 1.0, Nk, device=dev).view(1, Nk, 1)
        qpre = (q * (amp / 64.0)).to(BF16)              # base-2 logits grow to ~amp along the keys
        k = k.to(BF16)
        v = torch.randn(B, Nk, D, generator=g, device=dev).to(BF16)
        outs = []
        try:
            for mode in (0, 1):
                hip.call('crl_attn_fwd_set_mode', mode)
                o = torch.full_like(qpre, float('nan'))
                lse = torch.full((B, H, Nq), float('nan'), device=dev)
                ops.attn_fwd(qpre, k, v, o, lse, H, 0.125, False, q_prescaled=True)
                outs.append((o, lse))
        finally:
            hip.call('crl_attn_fwd_set_mode', 0)
        hd = lambda t: t.float().reshape(B, -1, H, 64).transpose(1, 2)
        z = hd(qpre) @ hd(k).transpose(-1, -2) * math.log(2.0)
        ref = (torch.softmax(z, -1) @ hd(v)).transpose(1, 2).reshape(B, Nq, D)
        assert torch.isfinite(outs[0][0].float()).all() and torch.isfinite(outs[0][1]).all()
        close(outs[0][0], ref, 2e-2, 2e-2, f'overflowing stream out, amp {amp}')
        assert float((outs[0][1] - torch.logsumexp(z, -1)).abs().max()) < 5e-3 * max(1.0, float(z.abs().max()) / 50)
        if must_equal_mode1:
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_gemm_column_scale(dev):
    """crl_gemm_bf16 colscale: the first colscale_cols columns of the plain bf16 epilogue are multiplied before the rounding -- through
    the 256x256 kernel, the 128x128 kernel, the remainder rows and the split-contraction reduce"""
    from pixparse_amd import hip, ops
    cs = 0.125 * ops.LOG2E
    for (M, N, K, cols) in [(256 * 5 + 72, 1536, 256, 512), (300, 384, 128, 128), (256 * 64 + 232, 1024, 2048, 512), (254, 768, 8192, 256),
                            (2048 + 72, 8192, 128, 4100 - 4)]:      # the last one: N >= 8192 takes the 256x128 two-per-CU kernel
        x = rnd((M, K), dev, 1.0, 1, BF16)
        w = rnd((N, K), dev, 0.05, 2, BF16)
        bias = rnd((N,), dev, 0.5, 3)
        out = torch.empty(M, N, dtype=BF16, device=dev)
        ops.linear_fwd(x, w, bias, out, colscale=cs, colscale_cols=cols)
        ref = x.float() @ w.float().t() + bias.to(BF16).float()
        ref[:, :cols] *= cs
        close(out, ref, 1e-2, 2e-2, f'colscale {M}x{N}x{K}')
        plain = torch.empty_like(out)
        ops.linear_fwd(x, w, bias, plain)
        assert torch.equal(out[:, cols:], plain[:, cols:])                 # the other columns are untouched, bit for bit
    with pytest.raises(hip.HipLibraryError):
        ops.linear_fwd(x, w, bias, torch.empty(M, N, device=dev), ops.EPI_F32_RESID, resid=torch.empty(M, N, device=dev), colscale=cs, colscale_cols=64)


# ------------------------------------------------------------------------------------------- attention backward, single-pass form
@pytest.mark.parametrize('B,H,Nq,Nk,chain', [(1, 2, 700, 1300, 4), (2, 4, 1023, 1200, 4), (1, 8, 577, 577, 2), (1, 2, 6189, 1100, 4), (1, 1, 300, 2049, 4)])
def test_attention_backward_query_split_of_the_remainder_chains(dev, B, H, Nq, Nk, chain):
    """crl_attn_bwd_set_qsplit(1): the key blocks a head has left over after its full chains are walked by two workgroups of half the query tiles
    each (the second half's dK / dV through a scratch behind the slabs + attn_bwd_addkv_kernel).  Against the unsplit launch at the same chain:
    dQ bit for bit (rows are disjoint, same slab), dK / dV bit for bit on the key rows of the full chains and within one bf16 rounding on the
    remainder's; twice the same bits; ragged key / query tails, strided views, B H = 8 (XCD order), a remainder of one and of two key blocks"""
    from pixparse_amd import hip, ops
    D, scale = H * 64, 0.125
    g = torch.Generator(device=dev).manual_seed(Nq + 3 * Nk)
    if Nq == Nk:
        qkv = torch.randn(B, Nq, 3 * D, generator=g, device=dev)
        qkv[:, :, :D] *= scale * ops.LOG2E
        qkv = qkv.to(BF16)
        q, k, v = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
    else:
        q = (torch.randn(B, Nq, D, generator=g, device=dev) * scale * ops.LOG2E).to(BF16)
        k, v = (torch.randn(B, Nk, D, generator=g, device=dev).to(BF16) for _ in range(2))
    do = torch.randn(B, Nq, D, generator=g, device=dev).to(BF16)
    o = torch.empty(B, Nq, D, dtype=BF16, device=dev)
    lse = torch.empty(B, H, Nq, device=dev)
    ops.attn_fwd(q, k, v, o, lse, H, scale, False, q_prescaled=True)
    delta = torch.empty(2, B, H, Nq, device=dev)

    def bwd(split):
        dq, dk, dv = (torch.full((B, n, D), float('nan'), dtype=BF16, device=dev) for n in (Nq, Nk, Nk))
        hip.call('crl_attn_bwd_set_mode', 2)
        hip.call('crl_attn_bwd_set_chain', chain)
        hip.call('crl_attn_bwd_set_qsplit', split)
        try:
            ops.attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, H, scale, False, q_prescaled=True)
        finally:
            hip.call('crl_attn_bwd_set_mode', 0)
            hip.call('crl_attn_bwd_set_chain', 0)
            hip.call('crl_attn_bwd_set_qsplit', -1)
        return dq, dk, dv
    ref, got, again = bwd(0), bwd(1), bwd(1)
    nkt = (Nk + 255) // 256
    assert nkt % chain != 0, 'the shape must leave a remainder chain'
    key_base = (nkt // chain) * chain * 256
    rel = lambda a, b: float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))
    assert torch.equal(got[0], ref[0]), 'dQ'
    for x, y, what in ((got[1], ref[1], 'dK'), (got[2], ref[2], 'dV')):
        assert torch.equal(x[:, :key_base], y[:, :key_base]), what + ' of the full chains'
        assert not torch.equal(x[:, key_base:], y[:, key_base:]) or Nq < 256, what + ': the split did not run'
        assert rel(x[:, key_base:], y[:, key_base:]) < 4e-3, (what, rel(x[:, key_base:], y[:, key_base:]))
    for x, y in zip(got, again):
        assert torch.equal(x, y), 'not reproducible'


@pytest.mark.parametrize('chain', [1, 4])
def test_attention_backward_persistent_launch_pulls_the_same_chains(dev, chain):
    """crl_attn_bwd_set_persistent: with more chains than CUs (here: all but 32 CUs reserved) the single pass is launched as one workgroup per CU that
    pulls its chains from the per-XCD ticket lists and steals at the end -- bit-identical to one workgroup per chain, with and without the query
    split, and to the static walk (crl_gemm_set_schedule(0)); forty launches through the slot ring leave the counters zeroed"""
    from pixparse_amd import hip, ops
    B, H, Nq, Nk = 4, 8, 577, 2600          # 11 key blocks x 32 heads: 352 chains of one, (2 + 2) x 32 = 128 items at chain 4 with the split, for 32 CUs
    D, scale = H * 64, 0.125
    g = torch.Generator(device=dev).manual_seed(7)
    q = (torch.randn(B, Nq, D, generator=g, device=dev) * scale * ops.LOG2E).to(BF16)
    k, v = (torch.randn(B, Nk, D, generator=g, device=dev).to(BF16) for _ in range(2))
    do = torch.randn(B, Nq, D, generator=g, device=dev).to(BF16)
    o = torch.empty(B, Nq, D, dtype=BF16, device=dev)
    lse = torch.empty(B, H, Nq, device=dev)
    ops.attn_fwd(q, k, v, o, lse, H, scale, False, q_prescaled=True)
    delta = torch.empty(2, B, H, Nq, device=dev)

    def bwd(persist, dynamic=True):
        dq, dk, dv = (torch.full((B, n, D), float('nan'), dtype=BF16, device=dev) for n in (Nq, Nk, Nk))
        hip.call('crl_attn_bwd_set_persistent', persist)
        ops.gemm_set_schedule(dynamic)
        ops.attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, H, scale, False, q_prescaled=True)
        return dq, dk, dv
    try:
        hip.call('crl_attn_bwd_set_mode', 2)
        hip.call('crl_attn_bwd_set_chain', chain)
        hip.call('crl_attn_bwd_set_qsplit', 1)
        ops.gemm_set_reserved_cus(224)
        want = bwd(0)
        for rep in range(40):
            got = bwd(1)
            if rep in (0, 39):
                for x, y in zip(got, want):
                    assert torch.equal(x, y), f'persistent launch differs (rep {rep})'
        for x, y in zip(bwd(1, dynamic=False), want):
            assert torch.equal(x, y), 'static walk differs'
    finally:
        ops.gemm_set_reserved_cus(0)
        ops.gemm_set_schedule(True)
        hip.call('crl_attn_bwd_set_persistent', 1)
        hip.call('crl_attn_bwd_set_qsplit', -1)
        hip.call('crl_attn_bwd_set_chain', 0)
        hip.call('crl_attn_bwd_set_mode', 0)


@pytest.mark.parametrize('B,H,Nq,Nk,pre', [(1, 2, 300, 700, True), (2, 1, 64, 512, True), (1, 2, 100, 45, True), (1, 2, 1023, 1300, True),
                                           (2, 2, 577, 577, True), (1, 1, 2100, 1100, True), (1, 2, 300, 700, False), (1, 1, 130, 260, False),
                                           (2, 4, 130, 600, True)])
def test_attention_backward_single_pass_mode(dev, B, H, Nq, Nk, pre):
    """crl_attn_bwd_set_mode(2): dK, dV and dQ from ONE recomputation of S / dP (5 MFMA products), dQ as a sum of per-key-block bf16 slabs
    reduced in fixed order.  pre: q prescaled = the hand-placed instruction stream (attn_bwd_spx_kernel), else the C++ form of the same
    algorithm; mode 3 runs the C++ form on the prescaled problem too and must agree with the stream BIT FOR BIT.  Against the two-pass
    form: dV bit for bit, dK to 1e-4, dQ within the extra bf16 rounding of the partials; against fp32 torch: the tolerance of the two-pass
    tests; two runs bit-identical.  Shapes cover a ragged last key block, fewer keys than one workgroup owns (256), ragged query tiles,
    more tiles than the ring / unroll period, strided q | k | v views and the cross-attention aspect ratio; the (1, 1, 2100, 1100) shape is
    long enough for mode 0 (auto) to pick the single pass by itself; the last one has B * H = 8 heads (the XCD-aware workgroup order)."""
    from pixparse_amd import hip, ops
    D, scale = H * 64, 0.125
    c = scale * ops.LOG2E if pre else 1.0
    g = torch.Generator(device=dev).manual_seed(Nq + Nk)
    if Nq == Nk:       # column blocks of one projection output, like the ViT blocks
        qkv = torch.randn(B, Nq, 3 * D, generator=g, device=dev)
        qkv[:, :, :D] *= c
        qkv = qkv.to(BF16)
        q, k, v = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
    else:
        q = (torch.randn(B, Nq, D, generator=g, device=dev) * c).to(BF16)
        k, v = (torch.randn(B, Nk, D, generator=g, device=dev).to(BF16) for _ in range(2))
    do = torch.randn(B, Nq, D, generator=g, device=dev).to(BF16)
    o = torch.empty(B, Nq, D, dtype=BF16, device=dev)
    lse = torch.empty(B, H, Nq, device=dev)
    ops.attn_fwd(q, k, v, o, lse, H, scale, False, q_prescaled=pre)
    delta = torch.empty(2, B, H, Nq, device=dev)

    def bwd(mode, chain=0):
        dq, dk, dv = (torch.full((B, n, D), float('nan'), dtype=BF16, device=dev) for n in (Nq, Nk, Nk))
        hip.call('crl_attn_bwd_set_mode', mode)
        hip.call('crl_attn_bwd_set_chain', chain)
        try:
            auto = Nq >= 1000 and Nk >= 1024
            assert (hip.query('crl_attn_bwd_ws_bytes', B, H, Nq, Nk, 0) > 0) == (mode >= 2 or (mode == 0 and auto))
            assert hip.query('crl_attn_bwd_ws_bytes', B, H, Nq, Nk, 1) == 0          # causal: always two-pass
            ops.attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, H, scale, False, q_prescaled=pre)
        finally:
            hip.call('crl_attn_bwd_set_mode', 0)
            hip.call('crl_attn_bwd_set_chain', 0)
        return dq, dk, dv
    dq2, dk2, dv2 = bwd(1)
    dq1, dk1, dv1 = bwd(2)
    rel = lambda a, b: float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))
    # dV bit for bit (P does not depend on the row constants); dK up to the summation order of delta = rowsum(dO o O), which the two
    # forms compute in different kernels (a last-bit difference in delta flips a bf16 rounding of dS here and there)
    assert torch.equal(dv1, dv2) and rel(dk1, dk2) < 1e-4
    assert rel(dq1, dq2) < 1e-2
    if pre:
        dq3, dk3, dv3 = bwd(3)
        assert torch.equal(dq3, dq1) and torch.equal(dk3, dk1) and torch.equal(dv3, dv1)
        dq0, dk0, dv0 = bwd(0)
        ref0 = (dq1, dk1, dv1) if (Nq >= 1000 and Nk >= 1024) else (dq2, dk2, dv2)
        assert torch.equal(dq0, ref0[0]) and torch.equal(dk0, ref0[1]) and torch.equal(dv0, ref0[2])
    hd = lambda t: t.float().reshape(B, -1, H, 64).transpose(1, 2)
    Qp, K, V = (hd(q) / c).requires_grad_(True), hd(k).requires_grad_(True), hd(v).requires_grad_(True)
    (torch.softmax(Qp @ K.transpose(-1, -2) * scale, -1) @ V).backward(hd(do))
    back = lambda t: t.transpose(1, 2).reshape(B, -1, D)
    assert rel(dq1, back(Qp.grad)) < 2e-2 and rel(dk1, back(K.grad)) < 2e-2 and rel(dv1, back(V.grad)) < 2e-2
    dq4, dk4, dv4 = bwd(2)
    assert torch.equal(dq4, dq1) and torch.equal(dk4, dk1) and torch.equal(dv4, dv1)
    # chains (crl_attn_bwd_set_chain): a workgroup walks several consecutive key blocks and adds each block's partial dQ to what the blocks
    # before it left in the slab -- fewer slabs for the reduce, one more bf16 rounding of the running sum per link.  dK / dV do not change
    # at all; the stream and its C++ form agree bit for bit at every chain length (including one longer than the key blocks there are, and
    # one that leaves a remainder chain); dQ stays within the single-pass tolerances.
    nkt = (Nk + 255) // 256
    dk_c1, dv_c1 = bwd(2, 1)[1:]
    for chain in sorted({1, 2, 3, nkt + 1}):
        dqc, dkc, dvc = bwd(2, chain)
        assert torch.equal(dkc, dk_c1) and torch.equal(dvc, dv_c1), chain
        assert rel(dqc, dq2) < 1e-2 and rel(dqc, back(Qp.grad)) < 2e-2, chain
        if pre:
            dqr, dkr, dvr = bwd(3, chain)
            assert torch.equal(dqr, dqc) and torch.equal(dkr, dkc) and torch.equal(dvr, dvc), chain
        if chain == 1 or nkt == 1:
            assert torch.equal(dqc, bwd(2, 1)[0])


# ------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize('M,N,K', [(300, 264, 192), (129, 128, 64), (1000, 520, 96), (64, 8, 32), (257, 1024, 1024)])
def test_gemm_nt_epilogues(dev, M, N, K):
    from pixparse_amd import ops
    x = rnd((M, K), dev, 1.0, 1, BF16)
    w = rnd((N, K), dev, 0.1, 2, BF16)
    bias = rnd((N,), dev, 0.5, 3)
    ref = x.float() @ w.float().t() + bias.to(BF16).float()
    out = torch.empty(M, N, dtype=BF16, device=dev)
    ops.linear_fwd(x, w, bias, out)
    close(out, ref, 1e-2, 1e-2, 'EPI_BF16')           # bf16 output rounding: 2^-8 relative
    # GELU epilogue: h = bf16(v + b), out = gelu(h), aux = gelu'(h) as fp16 (h itself may differ from torch's by a bf16 ulp: summation order)
    dact = torch.empty(M, N, dtype=F16, device=dev)
    act = torch.empty(M, N, dtype=BF16, device=dev)
    ops.linear_fwd(x, w, bias, act, ops.EPI_BF16_GELU, aux=dact)
    h = ref.to(BF16).float()
    close(dact, gelu_grad_ref(h), 1e-2, 1e-2, 'GELU aux')
    close(act, torch.nn.functional.gelu(h), 1e-2, 1e-2, 'GELU out')
    # ... and EXACTLY: with x = 0 the pre-activation is bf16(bias), known bit for bit -> gelu to a bf16 ulp, gelu' to an fp16 ulp, over [-9, 9]
    bsweep = torch.linspace(-9.0, 9.0, N, device=dev)
    ops.linear_fwd(torch.zeros_like(x), w, bsweep, act, ops.EPI_BF16_GELU, aux=dact)
    hb = bsweep.to(BF16).float().expand(M, N)
    close(act, torch.nn.functional.gelu(hb.double()).float(), 2.0 ** -8, 1e-6, 'GELU out (exact h)')
    close(dact, gelu_grad_ref(hb), 2.0 ** -10, 1e-6, 'GELU aux (exact h)')
    # residual epilogue (fp32 out, in place on the residual)
    resid = rnd((M, N), dev, 1.0, 4)
    want = resid + ref.to(BF16).float()
    y = resid.clone()
    ops.linear_fwd(x, w, bias, y, ops.EPI_F32_RESID, resid=y)
    close(y, want, 1e-2, 2e-2, 'F32_RESID')
    # no bias
    ops.linear_fwd(x, w, None, out)
    close(out, x.float() @ w.float().t(), 1e-2, 1e-2, 'no bias')


@pytest.mark.parametrize('M,N,K', [(1, 1024, 1024), (2, 264, 200), (7, 4096, 1024), (16, 1024, 4096), (3, 50304, 128), (5, 20, 40)])
def test_linear_skinny_vs_gemm_semantics(dev, M, N, K):
    """crl_linear_skinny_bf16 (decode-time projections): same arithmetic as the GEMM epilogues, strided output rows"""
    from pixparse_amd import ops
    x = rnd((M, K), dev, 1.0, 1, BF16)
    w = rnd((N, K), dev, 0.1, 2, BF16)
    bias = rnd((N,), dev, 0.5, 3)
    ref = x.float() @ w.float().t() + bias.to(BF16).float()
    big = torch.zeros(M, 3, N, dtype=BF16, device=dev)                # rows of a KV cache: out.stride(0) = 3N
    ops.linear_skinny(x, w, bias, big[:, 1, :])
    close(big[:, 1, :], ref, 1e-2, 1e-2, 'skinny EPI_BF16')
    assert float(big[:, 0, :].abs().max()) == 0 and float(big[:, 2, :].abs().max()) == 0
    act = torch.empty(M, N, dtype=BF16, device=dev)
    ops.linear_skinny(x, w, bias, act, ops.EPI_BF16_GELU)
    close(act, torch.nn.functional.gelu(ref.to(BF16).float()), 1e-2, 2e-3, 'skinny GELU')
    resid = rnd((M, N), dev, 1.0, 4)
    y = resid.clone()
    ops.linear_skinny(x, w, bias, y, ops.EPI_F32_RESID, resid=y)
    close(y, resid + ref.to(BF16).float(), 1e-2, 2e-2, 'skinny F32_RESID')
    out = torch.empty(M, N, dtype=BF16, device=dev)
    ops.linear_skinny(x, w, None, out)
    close(out, x.float() @ w.float().t(), 1e-2, 1e-2, 'skinny no bias')
    row = torch.tensor([2], dtype=torch.int32, device=dev)             # device-selected output row (KV-cache write of a graph step)
    big.zero_()
    ops.linear_skinny(x, w, bias, big[:, 0, :], out_row=row, out_row_stride=N)
    close(big[:, 2, :], ref, 1e-2, 1e-2, 'skinny device row')
    assert float(big[:, :2, :].abs().max()) == 0
    if M <= 16 and N >= 128 and K % 64 == 0:                           # bit-identical to the tiled GEMM? (different summation order: close)
        g = torch.empty(M, N, dtype=BF16, device=dev)
        ops.linear_fwd(x, w, None, g)
        close(out, g, 8e-3, 8e-3, 'skinny vs gemm')


@pytest.mark.parametrize('M,N,K', [(8, 3072, 1024), (1, 1024, 1024), (16, 4096, 768), (5, 784, 256)])
def test_linear_skinny_with_layernorm_in_front(dev, M, N, K):
    """crl_linear_skinny_ln_bf16 == crl_layernorm_fwd followed by crl_linear_skinny_bf16, bit for bit (projection output, GELU variant,
    device-selected output row, fp32 LayerNorm output), and both against fp32 torch"""
    from pixparse_amd import hip, ops
    t = rnd((M, K), dev, 2.0, 1) + 0.5
    gamma, beta = 1.0 + rnd((K,), dev, 0.2, 2), rnd((K,), dev, 0.3, 3)
    w = rnd((N, K), dev, 0.1, 4, BF16)
    bias = rnd((N,), dev, 0.5, 5)
    y32, y16 = torch.empty(M, K, device=dev), torch.empty(M, K, dtype=BF16, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    ops.layernorm_fwd(t, gamma, beta, 1e-5, y32, y16, mean, rstd)
    for epi in (ops.EPI_BF16, ops.EPI_BF16_GELU):
        want = torch.empty(M, N, dtype=BF16, device=dev)
        ops.linear_skinny(y16, w, bias, want, epi)
        got = torch.full((M, N), float('nan'), dtype=BF16, device=dev)
        h = torch.full((M, K), float('nan'), device=dev)
        ops.linear_skinny_ln(t, gamma, beta, 1e-5, h if N >= K else None, w, bias, got, epi)
        assert torch.equal(got, want), f'epi {epi}: {float((got.float() - want.float()).abs().max())}'
        if N >= K:
            assert torch.equal(h, y32)
    ref = torch.nn.functional.layer_norm(t, (K,), gamma, beta, 1e-5).to(BF16).float() @ w.float().t() + bias.to(BF16).float()
    out = torch.empty(M, N, dtype=BF16, device=dev)
    ops.linear_skinny_ln(t, gamma, beta, 1e-5, None, w, bias, out)
    close(out, ref, 2e-2, 2e-2, 'skinny + LN vs torch')
    big = torch.zeros(M, 3, N, dtype=BF16, device=dev)
    row = torch.tensor([2], dtype=torch.int32, device=dev)
    ops.linear_skinny_ln(t, gamma, beta, 1e-5, None, w, bias, big[:, 0, :], out_row=row, out_row_stride=N)
    assert torch.equal(big[:, 2, :], out) and float(big[:, :2, :].abs().max()) == 0
    if N < K:
        with pytest.raises(hip.HipLibraryError):
            ops.linear_skinny_ln(t, gamma, beta, 1e-5, torch.empty(M, K, device=dev), w, bias, out)


@pytest.mark.parametrize('M,N,K', [(300, 288, 192), (130, 96, 288), (1000, 544, 1024)])
def test_gemm_nn_dgrad(dev, M, N, K):
    """dx[M, K] = dy[M, N] @ w[N, K] (+ fused GELU backward)"""
    from pixparse_amd import ops
    dy = rnd((M, N), dev, 1.0, 1, BF16)
    w = rnd((N, K), dev, 0.1, 2, BF16)
    ref = dy.float() @ w.float()
    out = torch.empty(M, K, dtype=BF16, device=dev)
    ops.linear_dgrad(dy, w, out)
    close(out, ref, 1e-2, 1e-2, 'NN')
    h = rnd((M, K), dev, 1.0, 3, BF16)
    hf = h.float().requires_grad_(True)
    torch.nn.functional.gelu(hf).backward(ref.to(BF16).float())
    ops.linear_dgrad(dy, w, out, ops.EPI_BF16_DGELU, aux=gelu_grad_ref(h.float()).to(F16))      # aux = the derivative the forward saved
    close(out, hf.grad, 1e-2, 1e-2, 'NN + dGELU')
    acc = rnd((M, K), dev, 1.0, 5)
    want = acc + ref
    ops.linear_dgrad(dy, w, acc, ops.EPI_F32_ACC)
    close(acc, want, 1e-3, 1e-2, 'NN F32_ACC')


@pytest.mark.parametrize('M,N,K', [(777, 264, 192), (64, 128, 128), (1000, 520, 96), (4999, 256, 588)])
def test_gemm_tn_wgrad(dev, M, N, K):
    """dw[N, K] = dy[M, N]^T @ x[M, K]; contraction over the ragged row count M"""
    from pixparse_amd import ops
    dy = rnd((M, N), dev, 1.0, 1, BF16)
    Kp = (K + 63) // 64 * 64
    xs = torch.zeros(M, Kp, dtype=BF16, device=dev)
    xs[:, :K] = rnd((M, K), dev, 1.0, 2, BF16)
    ref = dy.float().t() @ xs[:, :K].float()
    dw = torch.full((N, K), 7.0, dtype=F32, device=dev)
    ops.linear_wgrad(dy, xs, dw, accumulate=False, k=K)
    close(dw, ref, 2e-3, 2e-2 * math.sqrt(M / 64), 'TN')     # fp32 accumulate of bf16 products
    ops.linear_wgrad(dy, xs, dw, accumulate=True, k=K)
    close(dw, 2 * ref, 2e-3, 4e-2 * math.sqrt(M / 64), 'TN accumulate')


def test_gemm_strided_views_and_errors(dev):
    from pixparse_amd import hip, ops
    M, D = 200, 128
    qkv = rnd((M, 3 * D), dev, 1.0, 1, BF16)
    w = rnd((64, D), dev, 0.1, 2, BF16)
    out = torch.empty(M, 64, dtype=BF16, device=dev)
    ops.linear_fwd(qkv[:, D:2 * D], w, None, out)            # row-strided A
    close(out, qkv[:, D:2 * D].float() @ w.float().t(), 1e-2, 1e-2, 'strided A')
    with pytest.raises(hip.HipLibraryError):
        ops.linear_fwd(rnd((8, 40), dev, 1, 1, BF16), rnd((8, 40), dev, 1, 2, BF16), None, torch.empty(8, 8, dtype=BF16, device=dev))  # K % 32


@pytest.mark.parametrize('policy,big', [(2, 1), (2, 0)])
@pytest.mark.parametrize('K', [64, 128, 192, 448])
def test_gemm_256_kernel_all_layouts(dev, K, policy, big):
    """the two 256x256 kernels (policy 2: big = 1 the 4-wave one-wave-per-SIMD kernel of round 5, big = 0 the 8-wave 8-phase kernel) forced on
    ragged shapes (odd / even / single K-tile counts, ragged M and N tiles), against fp32 torch; then the split-K wgrad path through fp32 slabs"""
    from pixparse_amd import hip, ops
    M, N = 600, 520
    hip.call('crl_gemm_set_big_kernel', big)
    x = rnd((M, K), dev, 1.0, 1, BF16)
    w = rnd((N, K), dev, 0.1, 2, BF16)
    bias = rnd((N,), dev, 0.5, 3)
    dy = rnd((M, N), dev, 1.0, 4, BF16)
    hip.call('crl_gemm_set_policy', policy)
    try:
        out = torch.empty(M, N, dtype=BF16, device=dev)
        pre = torch.empty(M, N, dtype=F16, device=dev)
        ops.linear_fwd(x, w, bias, out)
        close(out, x.float() @ w.float().t() + bias.to(BF16).float(), 1e-2, 1e-2, '256 NT bf16')
        ops.linear_fwd(x, w, bias, out, ops.EPI_BF16_GELU, aux=pre)
        ref = x.float() @ w.float().t() + bias.to(BF16).float()
        close(pre, gelu_grad_ref(ref.to(BF16).float()), 1e-2, 1e-2, '256 NT aux')
        close(out, torch.nn.functional.gelu(ref.to(BF16).float()), 1e-2, 1e-2, '256 NT gelu')
        y = rnd((M, N), dev, 1.0, 5)
        want = y + ref.to(BF16).float()
        ops.linear_fwd(x, w, bias, y, ops.EPI_F32_RESID, resid=y)
        close(y, want, 1e-2, 2e-2, '256 NT resid')
        if N % 32 == 8:                       # NN needs the contraction (N) to be a multiple of 32: use the first 512 columns
            dyc, wc = dy[:, :512], w[:512]
            dx = torch.empty(M, K, dtype=BF16, device=dev)
            ops.linear_dgrad(dyc, wc, dx)
            close(dx, dyc.float() @ wc.float(), 1e-2, 2e-2, '256 NN')
            h = rnd((M, K), dev, 1.0, 6, BF16)
            hf = h.float().requires_grad_(True)
            torch.nn.functional.gelu(hf).backward((dyc.float() @ wc.float()).to(BF16).float())
            ops.linear_dgrad(dyc, wc, dx, ops.EPI_BF16_DGELU, aux=gelu_grad_ref(h.float()).to(F16))
            close(dx, hf.grad, 1e-2, 2e-2, '256 NN + dGELU')
            accb = rnd((M, K), dev, 1.0, 7)
            wantb = accb + dyc.float() @ wc.float()
            ops.linear_dgrad(dyc, wc, accb, ops.EPI_F32_ACC)
            close(accb, wantb, 1e-3, 2e-2, '256 NN F32_ACC')
        dw = torch.full((N, K), 3.0, device=dev)
        ops.linear_wgrad(dy, x, dw, accumulate=True)
        close(dw, 3.0 + dy.float().t() @ x.float(), 2e-3, 5e-2, '256 TN acc')
    finally:
        hip.call('crl_gemm_set_policy', 0)
        hip.call('crl_gemm_set_big_kernel', 2)


@pytest.mark.parametrize('M,N,K', [(600, 520, 192), (256 * 9 + 40, 1024, 1024), (1024, 768, 64), (4096, 4096, 4096 + 64), (256 * 70 + 40, 1024, 512), (256 * 150, 512, 576)])
def test_gemm_4w_kernel_bit_identical_to_8w(dev, M, N, K):
    """gemm4w.hip (4 waves, one per SIMD, hand-placed main loop) against gemm256.hip (8 waves): every accumulator sums the same products in the
    same order and the epilogue arithmetic is the same, so all three layouts and all six epilogues agree BIT FOR BIT -- ragged M / N tiles, one /
    odd / even K-tile counts, persistent launches (more tiles than CUs) and split-K weight gradients included.  The plain-bf16 launches with whole
    column tiles and >= 8 K tiles run the OVERLAPPED form (the epilogue of tile T inside the main loop of tile T + 1, crl_gemm_set_overlap): one
    tile per workgroup (entry + drain statements only), several tiles per workgroup under the dynamic schedule (the last two shapes), and the
    classic form of the same kernel must all give the same bits"""
    from pixparse_amd import hip, ops
    x = rnd((M, K), dev, 1.0, 1, BF16)
    w = rnd((N, K), dev, 0.1, 2, BF16)
    bias = rnd((N,), dev, 0.5, 3)
    dy = rnd((M, N), dev, 1.0, 4, BF16)
    h = gelu_grad_ref(rnd((M, K), dev, 3.0, 6, BF16).float()).to(F16)       # a saved GELU derivative (fp16; pre-activations out to |h| ~ 12: fp16 subnormals and -0.0 included)
    y0 = rnd((M, N), dev, 1.0, 5)
    acc0 = rnd((M, K), dev, 1.0, 7)
    NC = N - N % 32                      # NN: the contraction (N) must be a multiple of 32
    KC = K - K % 64 if K % 64 else K
    CS = min(N, 328)                     # column-scale boundary INSIDE a 256-column tile (and not on a 16-column strip pair: 328 = 256 + 72)

    def run():
        out = torch.empty(M, N, dtype=BF16, device=dev); pre = torch.empty(M, N, dtype=F16, device=dev); g = torch.empty(M, N, dtype=BF16, device=dev)
        ops.linear_fwd(x, w, bias, out, colscale=0.25, colscale_cols=CS)
        ops.linear_fwd(x, w, bias, g, ops.EPI_BF16_GELU, aux=pre)
        y = y0.clone()
        ops.linear_fwd(x, w, bias, y, ops.EPI_F32_RESID, resid=y)
        dx = torch.empty(M, K, dtype=BF16, device=dev); dxg = torch.empty(M, K, dtype=BF16, device=dev)
        ops.linear_dgrad(dy[:, :NC], w[:NC], dx)
        ops.linear_dgrad(dy[:, :NC], w[:NC], dxg, ops.EPI_BF16_DGELU, aux=h)
        accb = acc0.clone()
        ops.linear_dgrad(dy[:, :NC], w[:NC], accb, ops.EPI_F32_ACC)
        dw = torch.full((N, K), 3.0, device=dev)
        ops.linear_wgrad(dy, x, dw, accumulate=True)
        dw2 = torch.empty(N, K, device=dev)
        ops.linear_wgrad(dy, x, dw2, accumulate=False)
        return out, pre, g, y, dx, dxg, accb, dw, dw2

    names = ['NT bf16 + colscale', 'NT gelu aux', 'NT gelu', 'NT resid', 'NN', 'NN dgelu', 'NN f32 acc', 'TN acc', 'TN store']
    hip.call('crl_gemm_set_policy', 2)
    try:
        hip.call('crl_gemm_set_big_kernel', 0)
        want = run()
        hip.call('crl_gemm_set_big_kernel', 1)
        hip.call('crl_gemm_set_overlap', 7)      # the overlapped form also for launches of fewer than three rounds of tiles
        got = run()
        again = run()
        hip.call('crl_gemm_set_overlap', 0)
        classic = run()
        hip.call('crl_gemm_set_overlap', 7)
        ops.gemm_set_schedule(False)
        static = run()
        ops.gemm_set_schedule(True)
    finally:
        hip.call('crl_gemm_set_overlap', 1)
        ops.gemm_set_schedule(True)
        hip.call('crl_gemm_set_policy', 0)
        hip.call('crl_gemm_set_big_kernel', 2)
    ref = x.float() @ w.float().t() + bias.to(BF16).float()
    ref[:, :CS] *= 0.25
    close(got[0], ref, 1e-2, 1e-2, '4w NT vs fp32')
    close(got[8], dy.float().t() @ x.float(), 2e-3, 0.2, '4w TN vs fp32')
    for n, a, b, c, d, e in zip(names, want, got, again, classic, static):
        assert torch.equal(a, b), f'{n}: 4-wave kernel differs from the 8-wave kernel (max abs {(a.float() - b.float()).abs().max().item():.3e})'
        assert torch.equal(b, c), f'{n}: 4-wave kernel is not reproducible'
        assert torch.equal(b, d), f'{n}: overlapped and classic epilogue differ'
        assert torch.equal(b, e), f'{n}: dynamic and static tile walk differ'


@pytest.mark.parametrize('big', [0, 1, 2])
def test_wgrad_with_a_partial_last_round_is_never_row_cut(dev, big):
    """a weight gradient whose output has more 256x256 tiles than CUs and a partial last round (71 x 4 tiles, contraction 2048: the shape class
    of the LM-head weight gradient, 197 x 4 tiles): the wave-quantisation cut is along the rows of a ROW-MAJOR A operand and must never be
    applied to the k-major weight-gradient layout (round 5 regression: an edit of quant_rows dropped that guard; the remainder launch then
    read A with the wrong strides -- illegal addresses at cfg-3, run-to-run differences in the loss)"""
    from pixparse_amd import hip, ops
    Kc, Mo, No = 2048, 256 * 70 + 40, 1024
    dy = rnd((Kc, Mo), dev, 1.0, 1, BF16)
    x = rnd((Kc, No), dev, 1.0, 2, BF16)
    hip.call('crl_gemm_set_big_kernel', big)
    try:
        dw = torch.full((Mo, No), 1.0, device=dev)
        ops.linear_wgrad(dy, x, dw, accumulate=True)
        dw2 = torch.full((Mo, No), 1.0, device=dev)
        ops.linear_wgrad(dy, x, dw2, accumulate=True)
    finally:
        hip.call('crl_gemm_set_big_kernel', 2)
    assert torch.equal(dw, dw2)
    close(dw, 1.0 + dy.float().t() @ x.float(), 2e-3, 0.5, 'TN, 284 tiles, partial last round')


def test_gemm_dynamic_schedule_per_stream_rings_and_graph_capture(dev):
    """the ticket-counter pool is a ring PER STREAM: persistent GEMMs in flight on two streams at once, and captured into a hipGraph
    (torch captures on a side stream: the captured launches keep their slots, replays are ordered), give the static walk's bits"""
    from pixparse_amd import hip, ops
    M, N, K = 256 * 41 + 72, 2048, 192
    x = rnd((M, K), dev, 1.0, 1, BF16)
    w = rnd((N, K), dev, 0.1, 2, BF16)
    bias = rnd((N,), dev, 0.5, 4)
    try:
        hip.call('crl_gemm_set_policy', 2)
        ops.gemm_set_schedule(False)
        want = torch.empty(M, N, dtype=BF16, device=dev)
        ops.linear_fwd(x, w, bias, want)
        ops.gemm_set_schedule(True)
        side = torch.cuda.Stream()
        outs_a = [torch.empty(M, N, dtype=BF16, device=dev) for _ in range(40)]
        outs_b = [torch.empty(M, N, dtype=BF16, device=dev) for _ in range(40)]
        torch.cuda.synchronize()
        for oa, ob in zip(outs_a, outs_b):               # 40 + 40 launches, the two streams unordered with respect to each other
            ops.linear_fwd(x, w, bias, oa)
            with torch.cuda.stream(side):
                ops.linear_fwd(x, w, bias, ob)
        torch.cuda.synchronize()
        for o in outs_a + outs_b:
            assert torch.equal(o, want), 'dynamic schedule on two concurrent streams differs from the static walk'
        og = torch.zeros(M, N, dtype=BF16, device=dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(3):
                ops.linear_fwd(x, w, bias, og)
        for _ in range(30):                              # 90 replayed launches through the capture stream's ring of 64
            g.replay()
            ops.linear_fwd(x, w, bias, outs_a[0])       # eager launches between the replays use the default stream's ring
        torch.cuda.synchronize()
        assert torch.equal(og, want) and torch.equal(outs_a[0], want), 'graph-captured dynamic schedule differs from the static walk'
    finally:
        hip.call('crl_gemm_set_policy', 0)
        ops.gemm_set_schedule(True)


def test_gemm_dynamic_tile_schedule(dev):
    """persistent GEMMs (more tiles than resident workgroups) under the dynamic tile scheduler: bit-identical to the static walk for all
    three layouts and both persistent kernels; the ticket counters are left zeroed by every launch (80 launches through a pool of 64
    slots); with CUs reserved for RCCL (crl_gemm_set_reserved_cus) and with CUs taken away behind the library's back by a sleeping
    side-stream kernel -- the workgroups that start late find the queue empty -- the results do not change"""
    from pixparse_amd import hip, ops
    M, N, K = 256 * 41 + 72, 2048, 192          # 42 x 8 = 336 tiles of 256 x 256 (672 of 256 x 128), ragged last row tile
    x = rnd((M, K), dev, 1.0, 1, BF16)
    w = rnd((N, K), dev, 0.1, 2, BF16)
    w2 = rnd((K, N), dev, 0.1, 3, BF16)
    bias = rnd((N,), dev, 0.5, 4)
    res = rnd((M, N), dev, 1.0, 5)

    def run():
        out = torch.empty(M, N, dtype=BF16, device=dev)
        ops.linear_fwd(x, w, bias, out)                                   # NT
        y = torch.empty(M, N, dtype=F32, device=dev)
        ops.linear_fwd(x, w, bias, y, ops.EPI_F32_RESID, resid=res)       # NT + fp32 residual
        o2 = torch.empty(M, N, dtype=BF16, device=dev)
        ops.linear_dgrad(x, w2, o2)                                       # NN
        return out, y, o2

    try:
        for policy in (2, 0):
            hip.call('crl_gemm_set_policy', policy)
            ops.gemm_set_schedule(False)
            want = run()
            ops.gemm_set_schedule(True)
            for rep in range(27):                                          # 81 persistent launches: the slot pool wraps
                got = run()
                if rep in (0, 26):
                    for g, w_ in zip(got, want):
                        assert torch.equal(g, w_), f'policy {policy} rep {rep}: dynamic schedule differs from the static walk'
            for reserved in (32, 100):
                ops.gemm_set_reserved_cus(reserved)
                for g, w_ in zip(run(), want):
                    assert torch.equal(g, w_), f'policy {policy}: result changed with {reserved} CUs reserved'
                ops.gemm_set_reserved_cus(0)
            with ops.OccupyCUs(48, max_seconds=20.0):
                for sched in (True, False):
                    ops.gemm_set_schedule(sched)
                    for g, w_ in zip(run(), want):
                        assert torch.equal(g, w_), f'policy {policy}: result changed with 48 CUs occupied (dynamic={sched})'
                torch.cuda.current_stream().synchronize()     # NOT a device-wide synchronize: that would wait for the sleepers
            ops.gemm_set_schedule(True)
        ref = x.float() @ w.float().t() + bias.to(BF16).float()
        close(want[0], ref, 1e-2, 1e-2, 'dynamic-schedule NT vs fp32')
        # wgrad layout: split-K launches keep the static one-tile-per-workgroup form; a wgrad with more output tiles than CUs
        # (17 x 20 tiles, short contraction) is a persistent launch and pulls tickets too
        dy = rnd((512, 4352), dev, 1.0, 6, BF16)
        xx = rnd((512, 5120), dev, 1.0, 7, BF16)
        hip.call('crl_gemm_set_policy', 2)
        dws = []
        for sched in (False, True):
            ops.gemm_set_schedule(sched)
            dw = torch.zeros(4352, 5120, device=dev)
            ops.linear_wgrad(dy, xx, dw, accumulate=False)
            dws.append(dw)
        assert torch.equal(dws[0], dws[1])
        close(dws[1], dy.float().t() @ xx.float(), 2e-3, 0.2, 'TN persistent')
    finally:
        hip.call('crl_gemm_set_policy', 0)
        ops.gemm_set_schedule(True)
        ops.gemm_set_reserved_cus(0)


def test_gemm_wave_quantisation_split(dev):
    """auto policy, 17 x 16 tiles of 256: the big kernel takes 16 row tiles (one full wave), the 128 kernel the last 104 rows (with so short
    a contraction the cost model would not cut: its remainder cost is zeroed for the test)"""
    from pixparse_amd import hip, ops
    M, N, K = 256 * 16 + 104, 4096, 128
    hip.call('crl_gemm_set_quant_cost', 0.0)
    try:
        _quant_split_case(dev, M, N, K)
    finally:
        hip.call('crl_gemm_set_quant_cost', 1.0)
    _quant_split_case(dev, M, N, K)          # and uncut (ragged last row tile inside the 256 kernel)


def _quant_split_case(dev, M, N, K):
    from pixparse_amd import ops
    x = rnd((M, K), dev, 1.0, 1, BF16)
    w = rnd((N, K), dev, 0.1, 2, BF16)
    bias = rnd((N,), dev, 0.5, 3)
    ref = x.float() @ w.float().t() + bias.to(BF16).float()
    pre = torch.empty(M, N, dtype=F16, device=dev)
    act = torch.empty(M, N, dtype=BF16, device=dev)
    ops.linear_fwd(x, w, bias, act, ops.EPI_BF16_GELU, aux=pre)
    close(pre, gelu_grad_ref(ref.to(BF16).float()), 1e-2, 1e-2, 'split aux')
    close(act, torch.nn.functional.gelu(ref.to(BF16).float()), 1e-2, 1e-2, 'split gelu')
    y = rnd((M, N), dev, 1.0, 4)
    want = y + ref.to(BF16).float()
    ops.linear_fwd(x, w, bias, y, ops.EPI_F32_RESID, resid=y)
    close(y, want, 1e-2, 2e-2, 'split resid')
    dy = rnd((M, K), dev, 1.0, 5, BF16)          # NN: out[M, N] = dy[M, K] @ w2[K, N]
    w2 = rnd((K, N), dev, 0.1, 6, BF16)
    out = torch.empty(M, N, dtype=BF16, device=dev)
    ops.linear_dgrad(dy, w2, out)
    close(out, dy.float() @ w2.float(), 1e-2, 1e-2, 'split NN')


def test_gemm_remainder_rows_split_contraction(dev):
    """auto policy with a long contraction (K = 2048 / 3072): the remainder rows of the wave-quantisation split (few 128-tiles) run as
    split-K slabs in the scratch + a reduce that applies the epilogue (bias + bf16, bias + bf16 rounding + fp32 residual)"""
    from pixparse_amd import hip, ops
    M, N = 256 * 64 + 232, 1024
    for K in (2048, 3072):
        assert hip.query('crl_gemm_ws_bytes', hip.NT, ops.EPI_F32_RESID, M, N, K) > 0
        assert hip.query('crl_gemm_ws_bytes', hip.NN, ops.EPI_BF16, M, N, K) > 0
        assert hip.query('crl_gemm_ws_bytes', hip.NT, ops.EPI_BF16_GELU, M, N, K) == 0
        x = rnd((M, K), dev, 1.0, 1, BF16)
        w = rnd((N, K), dev, 0.05, 2, BF16)
        bias = rnd((N,), dev, 0.5, 3)
        ref = x.float() @ w.float().t() + bias.to(BF16).float()
        out = torch.empty(M, N, dtype=BF16, device=dev)
        ops.linear_fwd(x, w, bias, out)
        close(out, ref, 1e-2, 2e-2, 'remainder split bf16')
        y = rnd((M, N), dev, 1.0, 4)
        want = y + ref.to(BF16).float()
        ops.linear_fwd(x, w, bias, y, ops.EPI_F32_RESID, resid=y)
        close(y, want, 1e-2, 3e-2, 'remainder split resid')
        w2 = rnd((K, N), dev, 0.05, 6, BF16)        # NN: out[M, N] = x[M, K] @ w2[K, N]
        ops.linear_dgrad(x, w2, out)
        close(out, x.float() @ w2.float(), 1e-2, 2e-2, 'remainder split NN')
        # round 4: the fp32 store / accumulate epilogues take the same route (the fused cross-attention K/V dgrad writes fp32)
        assert hip.query('crl_gemm_ws_bytes', hip.NN, hip.EPI_F32, M, N, K) > 0
        o32 = torch.full((M, N), 0.25, device=dev)
        ops.gemm(hip.NN, hip.EPI_F32, M, N, K, x, x.stride(0), w2, w2.stride(0), o32, N)
        close(o32, x.float() @ w2.float(), 2e-3, 2e-2, 'remainder split NN fp32')
        ops.gemm(hip.NN, hip.EPI_F32_ACC, M, N, K, x, x.stride(0), w2, w2.stride(0), o32, N)
        close(o32, 2 * (x.float() @ w2.float()), 2e-3, 4e-2, 'remainder split NN fp32 accumulate')


def test_gemm_few_tiles_long_contraction_split(dev):
    """a forward / dgrad GEMM with a handful of output tiles and a very long contraction (the LM-head dgrad of a small batch: 254 x 768 over
    K = 50304) is cut into up to 32 chunks of the contraction (fp32 slabs + the reduce that applies the epilogue)"""
    from pixparse_amd import hip, ops
    M, N, K = 254, 768, 50304
    assert hip.query('crl_gemm_ws_bytes', hip.NN, ops.EPI_BF16, M, N, K) == 32 * M * N * 4
    assert hip.query('crl_gemm_ws_bytes', hip.NN, ops.EPI_BF16, M, N, 4096) == 16 * M * N * 4   # 64 K tiles: 16 slabs of 4
    assert hip.query('crl_gemm_ws_bytes', hip.NN, ops.EPI_BF16, M, N, 512) == 0           # short contraction: one launch
    dy = rnd((M, K), dev, 1.0, 1, BF16)
    w = rnd((K, N), dev, 0.02, 2, BF16)
    out = torch.empty(M, N, dtype=BF16, device=dev)
    ops.linear_dgrad(dy, w, out)
    close(out, dy.float() @ w.float(), 1e-2, 3e-2, 'few-tiles split NN')
    x = rnd((M, 8192), dev, 1.0, 3, BF16)
    w2 = rnd((256, 8192), dev, 0.02, 4, BF16)
    bias = rnd((256,), dev, 0.5, 5)
    y = rnd((M, 256), dev, 1.0, 6)
    want = y + (x.float() @ w2.float().t() + bias.to(BF16).float()).to(BF16).float()
    ops.linear_fwd(x, w2, bias, y, ops.EPI_F32_RESID, resid=y)
    close(y, want, 1e-2, 3e-2, 'few-tiles split NT + residual')


def test_gemm_round_model_calibration(dev):
    """crl_gemm_calibrate: one / two rounds of 256x256 tiles timed at K = 1024 / 4096 on hashed bf16 operands refit the microseconds-per-round
    model of the wave-quantisation cut for THIS device.  Plausible numbers, and results of a cut GEMM unchanged."""
    from pixparse_amd import hip, ops
    seen = []
    for attempt in range(3):        # a timing: in the middle of the suite one attempt in a few lands outside (clock ramps of the launches before it)
        a, b, cal = ops.gemm_calibrate(dev, force=True)
        seen.append((a, b, cal))
        if cal and 0.0 <= a < 40.0 and 8.0 < b < 60.0:
            break
    assert cal and 0.0 <= a < 40.0 and 8.0 < b < 60.0, seen
    # one round of 256 tiles: ~20-50 us at K = 1024 on an MI355X (14-16 / 31-33 when run alone)
    M, N, K = 256 * 64 + 232, 1024, 2048
    x = rnd((M, K), dev, 1.0, 1, BF16)
    w = rnd((N, K), dev, 0.05, 2, BF16)
    out = torch.empty(M, N, dtype=BF16, device=dev)
    ops.linear_fwd(x, w, None, out)
    close(out, x.float() @ w.float().t(), 1e-2, 2e-2, 'GEMM after calibration')
    a2, b2, _ = ops.gemm_calibrate(dev)                                          # cached: no second measurement
    assert (a2, b2) == (a, b)


def test_gemm_quarter_full_chip_long_contraction_split(dev):
    """round 4: the few-tiles split of the 128x128 kernel also covers 128 ... 255 output tiles (a quarter to a half of the chip's 512
    workgroup slots) when the contraction is long (>= 64 K tiles): cfg-2's 4088-row decoder GEMMs are 192 tiles.  And it now serves the
    plain fp32 store / accumulate epilogues (dgrads into an fp32 gradient) next to bias + bf16 and bias + residual."""
    from pixparse_amd import hip, ops
    M, N, K = 2000, 1536, 4096                     # 16 x 12 = 192 tiles (48 tiles of 256^2 are too few for the big kernel), 64 K tiles -> 2 slabs
    assert hip.query('crl_gemm_ws_bytes', hip.NN, ops.EPI_BF16, M, N, K) == 2 * M * N * 4
    assert hip.query('crl_gemm_ws_bytes', hip.NN, ops.EPI_BF16, M, N, 2048) == 0             # 32 K tiles: one launch
    assert hip.query('crl_gemm_ws_bytes', hip.NN, hip.EPI_F32, 254, 768, 50304) == 32 * 254 * 768 * 4
    dy = rnd((M, K), dev, 1.0, 7, BF16)
    w = rnd((K, N), dev, 0.02, 8, BF16)
    ref = dy.float() @ w.float()
    out = torch.empty(M, N, dtype=BF16, device=dev)
    ops.linear_dgrad(dy, w, out)
    close(out, ref, 1e-2, 3e-2, '192-tile split NN bf16')
    o32 = torch.full((M, N), 0.5, device=dev)
    ops.gemm(hip.NN, hip.EPI_F32, M, N, K, dy, dy.stride(0), w, w.stride(0), o32, N)
    close(o32, ref, 2e-3, 2e-2, '192-tile split NN fp32')
    ops.gemm(hip.NN, hip.EPI_F32_ACC, M, N, K, dy, dy.stride(0), w, w.stride(0), o32, N)
    close(o32, 2 * ref, 2e-3, 4e-2, '192-tile split NN fp32 accumulate')
    o32b = torch.full((M, N), 0.5, device=dev)
    ops.gemm(hip.NN, hip.EPI_F32, M, N, K, dy, dy.stride(0), w, w.stride(0), o32b, N)
    ops.gemm(hip.NN, hip.EPI_F32_ACC, M, N, K, dy, dy.stride(0), w, w.stride(0), o32b, N)
    assert torch.equal(o32, o32b), 'slab reduction must be deterministic'


def test_gemm_half_chip_long_contraction_split(dev):
    """round 6: forward / dgrad GEMMs whose 256x256 tiles fill at most half the chip behind a contraction of >= 64 K tiles (the decoder's 8184-row
    fc2 / fc1 dgrad: 128 tiles x 64 K tiles; the LM-head dgrad: 128 tiles x 786) run as contraction slices of the 4-wave 256x256 kernel + the slab
    reduce that applies the epilogue, not as 512 small tiles: workspace, values against fp32, determinism, and the reservation of CUs switches it off"""
    from pixparse_amd import hip, ops
    M, N = 8184, 1024
    for K, tol in ((4096, 3e-2), (50304, 1e-1)):
        assert hip.query('crl_gemm_ws_bytes', hip.NN, ops.EPI_BF16, M, N, K) == 2 * M * N * 4
        dy = rnd((M, K), dev, 1.0, 7, BF16)
        w = rnd((K, N), dev, 0.02, 8, BF16)
        rows = torch.cat([torch.randperm(M, generator=torch.Generator().manual_seed(2))[:126], torch.tensor([0, M - 1])]).to(dev)
        ref = dy[rows].float() @ w.float()
        out = torch.empty(M, N, dtype=BF16, device=dev)
        ops.linear_dgrad(dy, w, out)
        close(out[rows], ref, 1e-2, tol, f'half-chip split NN bf16 K={K}')
        out2 = torch.empty(M, N, dtype=BF16, device=dev)
        ops.linear_dgrad(dy, w, out2)
        assert torch.equal(out, out2), 'slab reduction must be deterministic'
        hip.call('crl_gemm_set_policy', 1)                   # the 128x128 kernel: same values up to the summation order
        try:
            ops.linear_dgrad(dy, w, out2)
        finally:
            hip.call('crl_gemm_set_policy', 0)
        close(out[rows], out2[rows].float(), 1e-2, tol, 'against the small-tile kernel')
    # forward layout with bias + fp32 residual (fc2 of a decoder layer)
    K = 4096
    x = rnd((M, K), dev, 1.0, 1, BF16)
    wt = rnd((N, K), dev, 0.02, 2, BF16)
    bias = rnd((N,), dev, 0.5, 3)
    resid = rnd((M, N), dev, 1.0, 4)
    assert hip.query('crl_gemm_ws_bytes', hip.NT, ops.EPI_F32_RESID, M, N, K) == 2 * M * N * 4
    want = resid + (x.float() @ wt.float().t() + bias.to(BF16).float()).to(BF16).float()
    y = resid.clone()
    ops.linear_fwd(x, wt, bias, y, ops.EPI_F32_RESID, resid=y)
    close(y, want, 1e-2, 3e-2, 'half-chip split NT resid')
    ops.gemm_set_reserved_cus(16)                            # 240 CUs: 128 tiles are more than half -> one launch of small tiles again
    try:
        assert hip.query('crl_gemm_ws_bytes', hip.NT, ops.EPI_F32_RESID, M, N, K) == 0
    finally:
        ops.gemm_set_reserved_cus(0)


@pytest.mark.parametrize('Mrows,N,K', [(5000, 1024, 1024), (49512 // 4, 1024, 512), (6000, 520, 448), (300, 264, 192), (4000, 2048, 256), (70000, 256, 256),
                                       (2048, 4352, 5120), (8000, 512, 768)])      # the last two: a persistent launch (340 tiles, no contraction split), three column tiles (two share the duty)
def test_wgrad_with_bias_gradient(dev, Mrows, N, K):
    """linear_wgrad(..., dbias=...): the bias gradient (column sums of dy) from the weight-gradient call.  The 4-wave kernel sums the columns of its A
    operand on the matrix pipe (eight MFMAs against a fragment of ones on the K tiles a wave is on duty for; the two waves and the <= 4 column
    tiles that read the same A panel share the contraction) -- exact fp32 sums of the bf16 values like the stand-alone pass, in another order;
    every other kernel choice falls back to that pass inside the same entry point.  Against fp64 torch; accumulate and overwrite; twice the same
    bits; the weight gradient itself bit-identical to the call without dbias; ragged contraction (rows not a multiple of 64) and ragged N"""
    from pixparse_amd import hip, ops
    dy = rnd((Mrows, N), dev, 1.0, 1, BF16)
    x = rnd((Mrows, K), dev, 1.0, 2, BF16)
    ref = dy.double().sum(0)
    tol = 2e-5 * float(dy.float().abs().sum(0).max()) + 1e-4
    for big in (2, 1, 0):                      # per launch (product), the 4-wave kernel, the 8-wave kernel (stand-alone pass inside)
        hip.call('crl_gemm_set_big_kernel', big)
        try:
            dw0 = torch.zeros(N, K, device=dev)
            ops.linear_wgrad(dy, x, dw0, accumulate=False)
            dw = torch.zeros(N, K, device=dev)
            db = torch.full((N,), float('nan'), device=dev)
            ops.linear_wgrad(dy, x, dw, accumulate=False, dbias=db, dbias_accumulate=False)
            assert torch.equal(dw, dw0), f'big={big}: the weight gradient changed with the column sums on'
            assert float((db.double() - ref).abs().max()) < tol, (big, float((db.double() - ref).abs().max()), tol)
            db2 = torch.full((N,), 3.0, device=dev)
            ops.linear_wgrad(dy, x, dw, accumulate=True, dbias=db2, dbias_accumulate=True)
            assert float((db2.double() - 3.0 - ref).abs().max()) < tol
            db3 = torch.empty(N, device=dev)
            ops.linear_wgrad(dy, x, dw, accumulate=False, dbias=db3, dbias_accumulate=False)
            assert torch.equal(db3, db), f'big={big}: not reproducible'
        finally:
            hip.call('crl_gemm_set_big_kernel', 2)
    # a column-restricted call (n < dy.shape[1]: padded projections) sums only those columns
    if N >= 512:
        db = torch.empty(N - 256, device=dev)
        dw = torch.zeros(N - 256, K, device=dev)
        ops.linear_wgrad(dy, x, dw, accumulate=False, n=N - 256, dbias=db, dbias_accumulate=False)
        assert float((db.double() - ref[:N - 256]).abs().max()) < tol


def test_gemm_256_splitk_wgrad(dev):
    from pixparse_amd import hip, ops
    Mrows, N, K = 5000, 520, 448                # contraction over 5000 rows -> 79 K tiles, split into slabs
    dy = rnd((Mrows, N), dev, 1.0, 1, BF16)
    x = rnd((Mrows, K), dev, 1.0, 2, BF16)
    ref = dy.float().t() @ x.float()
    for policy in (2, 1, 0):
        hip.call('crl_gemm_set_policy', policy)
        try:
            assert hip.query('crl_gemm_ws_bytes', hip.TN, hip.EPI_F32_ACC, N, K, Mrows) > 0
            dw = torch.full((N, K), 1.0, device=dev)
            ops.linear_wgrad(dy, x, dw, accumulate=True)
            close(dw, 1.0 + ref, 2e-3, 0.2, f'split-K wgrad policy {policy}')
            dw2 = torch.full((N, K), 1.0, device=dev)
            ops.linear_wgrad(dy, x, dw2, accumulate=True)
            assert torch.equal(dw, dw2), 'slab reduction must be deterministic'
        finally:
            hip.call('crl_gemm_set_policy', 0)


def test_attention_backward_parts_hook(dev):
    """crl_attn_bwd_set_parts (measurement hook of bench.py / scripts: which launches of the TWO-PASS backward are issued): the dK / dV pass alone
    and the dQ pass alone write what the full call writes and leave the other outputs untouched; any mask but 7 keeps the single pass out"""
    from pixparse_amd import hip, ops
    B, H, Nq, Nk, scale = 1, 2, 300, 520, 0.125
    D = H * 64
    g = torch.Generator(device=dev).manual_seed(5)
    q, k, v, do = (torch.randn(B, n, D, generator=g, device=dev).to(BF16) for n in (Nq, Nk, Nk, Nq))
    o = torch.empty_like(q)
    lse = torch.empty(B, H, Nq, device=dev)
    ops.attn_fwd(q, k, v, o, lse, H, scale, False)

    def bwd(mask):
        dq, dk, dv = (torch.full((B, n, D), 7.0, dtype=BF16, device=dev) for n in (Nq, Nk, Nk))
        delta = torch.empty(2, B, H, Nq, device=dev)
        hip.call('crl_attn_bwd_set_parts', mask)
        try:
            ops.attn_bwd(q, k, v, o, do, lse, delta, dq, dk, dv, H, scale, False)
        finally:
            hip.call('crl_attn_bwd_set_parts', 7)
        return dq, dk, dv
    full = bwd(7)
    kv_only = bwd(1 | 2)
    q_only = bwd(1 | 4)
    rel = lambda a, b: float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))
    # (the partial runs take their row constants from the separate delta launch, the full call from the dQ pass's by-product: same values up to summation order)
    assert rel(kv_only[1], full[1]) < 2e-3 and rel(kv_only[2], full[2]) < 2e-3 and bool((kv_only[0] == 7.0).all())
    assert rel(q_only[0], full[0]) < 2e-3 and bool((q_only[1] == 7.0).all()) and bool((q_only[2] == 7.0).all())
    with pytest.raises(hip.HipLibraryError):
        hip.call('crl_attn_bwd_set_parts', 0)


# ------------------------------------------------------------------------------------------- LayerNorm & row ops
@pytest.mark.parametrize('M,D', [(37, 96), (1001, 1024), (5, 1536)])
def test_layernorm(dev, M, D):
    from pixparse_amd import ops
    x = rnd((M, D), dev, 2.0, 1) + 0.5
    gamma, beta = rnd((D,), dev, 0.2, 2) + 1.0, rnd((D,), dev, 0.2, 3)
    y32 = torch.empty(M, D, device=dev)
    y16 = torch.empty(M, D, dtype=BF16, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    ops.layernorm_fwd(x, gamma, beta, 1e-5, y32, y16, mean, rstd)
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-5)
    close(y32, ref, 1e-5, 1e-5, 'ln fwd f32')
    close(y16, ref, 1e-2, 1e-2, 'ln fwd bf16')
    dy32 = rnd((M, D), dev, 1.0, 4)
    dy16 = rnd((M, D), dev, 1.0, 5, BF16)
    ref.backward(dy32 + dy16.float())
    base = rnd((M, D), dev, 1.0, 6)
    dx = base.clone()
    dxb = torch.empty(M, D, dtype=BF16, device=dev)
    dg, db = torch.ones(D, device=dev), torch.ones(D, device=dev)
    dcol = torch.full((D,), 2.0, device=dev)
    ops.layernorm_bwd(dy32, dy16, x, gamma, mean, rstd, dx, True, dxb, dg, db, True, dx_colsum=dcol)
    close(dx, base + xr.grad, 1e-4, 1e-4, 'ln bwd dx (accumulated)')
    close(dxb, base + xr.grad, 1e-2, 1e-2, 'ln bwd dx bf16 = bf16(final)')
    close(dg, 1 + gr.grad, 1e-4, 1e-3, 'ln dgamma (+=)')
    close(db, 1 + br.grad, 1e-4, 1e-3, 'ln dbeta (+=)')
    # fused bias gradient of the Linear that produced x's branch: column sums of the bf16 tensor just written (+=)
    close(dcol, 2.0 + dxb.float().sum(0), 1e-5, 1e-3, 'ln bwd fused column sums of dx_bf16')
    dx_b, dxb_b, dg_b, db_b = base.clone(), torch.empty_like(dxb), torch.ones(D, device=dev), torch.ones(D, device=dev)
    ops.layernorm_bwd(dy32, dy16, x, gamma, mean, rstd, dx_b, True, dxb_b, dg_b, db_b, True)     # without: identical outputs
    assert torch.equal(dx, dx_b) and torch.equal(dxb, dxb_b) and torch.equal(dg, dg_b) and torch.equal(db, db_b)
    # in-place: dy_f32 aliases dx_f32, no accumulate
    d2 = dy32.clone()
    ops.layernorm_bwd(d2, None, x, gamma, mean, rstd, d2, False, None, None, None, True)
    xr2 = x.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr2, (D,), gamma, beta, 1e-5).backward(dy32)
    close(d2, xr2.grad, 1e-4, 1e-4, 'ln bwd in place')


def test_colsum_cast_add(dev):
    from pixparse_amd import ops
    x = rnd((1234, 264), dev, 1.0, 1, BF16)
    out = torch.ones(264, device=dev)
    ops.colsum(x, out, True)
    close(out, 1 + x.float().sum(0), 1e-4, 1e-2, 'colsum')
    s = rnd((1001,), dev, 3.0, 2)
    d = torch.empty(1001, dtype=BF16, device=dev)
    ops.cast_bf16(s, d)
    assert torch.equal(d, s.to(BF16))
    w = rnd((5, 7), dev, 1.0, 3)
    dp = torch.full((5, 16), 3.0, dtype=BF16, device=dev)
    ops.cast_pad_bf16(w, dp, 5, 7, 16)
    assert torch.equal(dp[:, :7], w.to(BF16)) and float(dp[:, 7:].abs().max()) == 0.0
    y = torch.ones(1001, device=dev)
    ops.add_bf16_to_f32(d, y, True)
    assert torch.equal(y, 1 + d.float())


def test_embed_scatter_is_deterministic_and_ordered(dev):
    """token-embedding gradient: duplicates (every sequence starts with the prompt token, padding occurs thousands of times)
    are added in position order on top of what the buffer holds (the tied LM-head gradient) -- bit-identical from run to
    run and equal to a sequential fp32 loop; runs longer than the 32-row chunk go through the partial-sum path"""
    from pixparse_amd import ops
    B, T, D, V = 4, 300, 128, 1000
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(3, V, (B, T), generator=g)
    ids[:, 0] = V - 1                                    # the prompt token of every sequence
    ids[:, 200:] = 1                                     # 400 padding rows: a run far longer than one chunk
    ids[1, 5:45] = 7                                     # a run of 40: two chunks
    ids = ids.to(dev)
    dt = rnd((B * T, D), dev, 1.0, 1)
    base = rnd((V, D), dev, 1.0, 2)                      # non-zero destination: (g + a) + b must not become (g + b) + a
    outs = []
    for _ in range(3):
        dtok, dpos = base.clone(), torch.zeros(T + 2, D, device=dev)
        ops.embed_bwd(ids, dt, dtok, dpos)
        outs.append(dtok)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    flat = ids.view(-1).cpu()
    ref64 = base.double().cpu().index_add_(0, flat, dt.double().cpu())
    close(outs[0], ref64.float().to(dev), 1e-5, 1e-4, 'embed scatter vs fp64 index_add')
    # exact order for short runs: the prompt token's row = base + (((dt[0] + dt[T]) + dt[2T]) + dt[3T]) in fp32, position order
    run = torch.zeros(D, device=dev)
    for b in range(B):
        run = run + dt[b * T]
    assert torch.equal(outs[0][V - 1], base[V - 1] + run)
    untouched = torch.ones(V, dtype=torch.bool)
    untouched[flat] = False
    assert torch.equal(outs[0][untouched.to(dev)], base[untouched.to(dev)])


def test_embed_tokens_im2row_merge(dev):
    from pixparse_amd import ops
    B, T, D, V = 3, 17, 64, 99
    ids = torch.randint(0, V, (B, T), device=dev)
    tok, pos = rnd((V, D), dev, 1, 1), rnd((T + 2, D), dev, 1, 2)
    out = torch.empty(B * T, D, device=dev)
    ops.embed_fwd(ids, tok, pos, out)
    assert torch.equal(out.view(B, T, D), tok[ids] + pos[torch.arange(T, device=dev) + 2])
    dt = rnd((B * T, D), dev, 1, 3)
    dtok, dpos = torch.zeros(V, D, device=dev), torch.ones(T + 2, D, device=dev)
    ops.embed_bwd(ids, dt, dtok, dpos)
    ref = torch.zeros(V, D, device=dev).index_add_(0, ids.view(-1), dt)
    close(dtok, ref, 1e-5, 1e-5, 'embed scatter')
    close(dpos[2:], 1 + dt.view(B, T, D).sum(0), 1e-5, 1e-5, 'embed dpos')
    # ViT token assembly
    Np = 6
    patch = rnd((B * Np, D), dev, 1, 4, BF16)
    cls, pe = rnd((1, 1, D), dev, 1, 5), rnd((1, Np + 1, D), dev, 1, 6)
    x = torch.empty(B * (Np + 1), D, device=dev)
    ops.vit_tokens_fwd(patch, cls, pe, x, B, Np, D)
    want = torch.cat([cls.expand(B, -1, -1), patch.view(B, Np, D).float()], 1) + pe
    assert torch.equal(x.view(B, Np + 1, D), want)
    dx = rnd((B * (Np + 1), D), dev, 1, 7)
    dpatch = torch.empty(B * Np, D, dtype=BF16, device=dev)
    dcls, dpe = torch.zeros(1, 1, D, device=dev), torch.zeros(1, Np + 1, D, device=dev)
    ops.vit_tokens_bwd(dx, dpatch, dcls, dpe, B, Np, D, True)
    d3 = dx.view(B, Np + 1, D)
    assert torch.equal(dpatch.view(B, Np, D), d3[:, 1:].to(BF16))
    close(dpe[0], d3.sum(0), 1e-5, 1e-5, 'dpos')
    close(dcls[0, 0], d3[:, 0].sum(0), 1e-5, 1e-5, 'dcls')
    # im2row == unfold of the floor-cropped image, column order (c, ph, pw)
    C, H, W, P = 3, 37, 50, 8
    img = rnd((B, C, H, W), dev, 1, 8)
    gh, gw = H // P, W // P
    Kp = 256
    patches = torch.empty(B * gh * gw, Kp, dtype=BF16, device=dev)
    ops.im2row(img, patches, P, gh, gw)
    ref = torch.nn.functional.unfold(img[:, :, :gh * P, :gw * P], P, stride=P).transpose(1, 2).reshape(B * gh * gw, C * P * P)
    assert torch.equal(patches[:, :C * P * P], ref.to(BF16)) and float(patches[:, C * P * P:].abs().max()) == 0
    # patch merge permutation
    Hf, Wf, Cc = 6, 4, 8
    xm = rnd((B, Hf, Wf, Cc), dev, 1, 9)
    ym = torch.empty(B, Hf // 2, Wf // 2, 4 * Cc, device=dev)
    ops.patch_merge_fwd(xm, ym, B, Hf, Wf, Cc)
    want = xm.reshape(B, Hf // 2, 2, Wf // 2, 2, Cc).permute(0, 1, 3, 4, 2, 5).flatten(3)
    assert torch.equal(ym, want)
    back = torch.empty_like(xm)
    ops.patch_merge_bwd(ym, back, B, Hf, Wf, Cc)
    assert torch.equal(back, xm)


# ------------------------------------------------------------------------------------------- attention
def attn_ref(q, k, v, scale, causal):
    """[B,H,N,d] fp32 restatement of the flash kernel: fp32 scores/softmax, P rounded to bf16 before P.V"""
    s = q @ k.transpose(-1, -2) * scale
    if causal:
        nq, nk = s.shape[-2:]
        s = s.masked_fill(~torch.ones(nq, nk, dtype=torch.bool, device=q.device).tril(nk - nq), float('-inf'))
    p = torch.softmax(s, -1)
    return p.to(BF16).float() @ v, torch.logsumexp(s, -1)


@pytest.mark.parametrize('B,H,Nq,Nk,causal', [(2, 2, 200, 200, False), (1, 3, 127, 333, False), (2, 2, 255, 255, True),
                                               (1, 8, 64, 64, False), (1, 1, 1, 70, False), (1, 2, 300, 300, True),
                                               (1, 2, 700, 700, False), (1, 1, 530, 530, True), (1, 2, 300, 900, False),
                                               (1, 1, 257, 449, True)])
def test_attention_fwd_bwd(dev, B, H, Nq, Nk, causal):
    from pixparse_amd import ops
    d, D = 64, H * 64
    scale = d ** -0.5
    # fused qkv buffer for self-attention shapes (strided views), separate tensors otherwise
    if Nq == Nk:
        qkv = rnd((B, Nq, 3 * D), dev, 1.0, 1, BF16)
        q, k, v = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
    else:
        q = rnd((B, Nq, D), dev, 1.0, 1, BF16)
        kv = rnd((B, Nk, 2 * D), dev, 1.0, 2, BF16)
        k, v = kv[:, :, :D], kv[:, :, D:]
    o = torch.empty(B, Nq, D, dtype=BF16, device=dev)
    lse = torch.empty(B, H, Nq, device=dev)
    ops.attn_fwd(q, k, v, o, lse, H, scale, causal)
    hd = lambda t, n: t.float().reshape(B, n, H, d).transpose(1, 2)
    qf, kf, vf = (hd(q, Nq).requires_grad_(True), hd(k, Nk).requires_grad_(True), hd(v, Nk).requires_grad_(True))
    oref, lref = attn_ref(qf, kf, vf, scale, causal)
    close(hd(o, Nq), oref, 2e-2, 2e-2, 'attn out')            # bf16 P and bf16 output
    close(lse, lref, 1e-3, 2e-3, 'attn lse')
    d_o = rnd((B, Nq, D), dev, 1.0, 5, BF16)
    oref.backward(hd(d_o, Nq))
    delta = torch.empty(2, B, H, Nq, device=dev)
    if Nq == Nk:
        dqkv = torch.zeros(B, Nq, 3 * D, dtype=BF16, device=dev)
        dq, dk, dv = dqkv[:, :, :D], dqkv[:, :, D:2 * D], dqkv[:, :, 2 * D:]
    else:
        dq = torch.zeros(B, Nq, D, dtype=BF16, device=dev)
        dkv = torch.zeros(B, Nk, 2 * D, dtype=BF16, device=dev)
        dk, dv = dkv[:, :, :D], dkv[:, :, D:]
    ops.attn_bwd(q, k, v, o, d_o, lse, delta, dq, dk, dv, H, scale, causal)
    # gradients: bf16 storage of sums of ~N products of bf16-rounded factors
    tol = 3e-2
    close(hd(dq, Nq), qf.grad, tol, tol * float(qf.grad.abs().max()), 'dq')
    close(hd(dk, Nk), kf.grad, tol, tol * float(kf.grad.abs().max()), 'dk')
    close(hd(dv, Nk), vf.grad, tol, tol * float(vf.grad.abs().max()), 'dv')


@pytest.mark.parametrize('B,H,Nk', [(1, 1, 1), (2, 3, 63), (2, 2, 64), (3, 2, 70), (2, 4, 1000), (1, 16, 6189), (8, 16, 300)])
def test_attention_decode_single_query(dev, B, H, Nk):
    """crl_attn_decode (split-KV single-query attention over a strided KV cache) vs the fp32 restatement and vs crl_attn_fwd"""
    from pixparse_amd import ops
    D = H * 64
    q = rnd((B, D), dev, 1.0, 1, BF16)
    cache = rnd((B, Nk + 5, 2 * D), dev, 1.0, 2, BF16)            # k | v per row, longer than the valid prefix
    k, v = cache[:, :Nk, :D], cache[:, :Nk, D:]
    o = torch.zeros(B, D, dtype=BF16, device=dev)
    ops.attn_decode(q, k, v, o, H, 0.125)
    hd = lambda t, n: t.float().reshape(B, n, H, 64).transpose(1, 2)
    oref, _ = attn_ref(hd(q[:, None], 1), hd(k, Nk), hd(v, Nk), 0.125, False)
    close(hd(o[:, None], 1), oref, 2e-2, 2e-2, 'attn decode')
    o2 = torch.empty(B, 1, D, dtype=BF16, device=dev)
    lse = torch.empty(B, H, 1, device=dev)
    ops.attn_fwd(q[:, None], k, v, o2, lse, H, 0.125, False)
    close(o, o2[:, 0], 1e-2, 1e-2, 'attn decode vs attn_fwd')
    # valid prefix taken from a device counter, the view spans the whole cache
    o3 = torch.zeros_like(o)
    ops.attn_decode(q, cache[:, :, :D], cache[:, :, D:], o3, H, 0.125, nk_minus1=torch.tensor([Nk - 1], dtype=torch.int32, device=dev))
    close(o3, o, 1e-2, 1e-2, 'attn decode, device key count')
    # query taken from a device-selected row of a [B, rows, 3D] cache
    qc = torch.zeros(B, 4, 3 * D, dtype=BF16, device=dev)
    qc[:, 3, :D] = q
    o4 = torch.zeros_like(o)
    ops.attn_decode(qc[:, 0, :D], k, v, o4, H, 0.125, q_row=torch.tensor([3], dtype=torch.int32, device=dev), q_row_stride=3 * D)
    assert torch.equal(o4, o)


def test_attention_forced_rescale(dev):
    """online-softmax rescale branch: one key far larger than the rest, placed in a late tile (guide rule 26)"""
    from pixparse_amd import ops
    B, H, N = 1, 1, 256
    q = rnd((B, N, 64), dev, 1.0, 1, BF16)
    k = rnd((B, N, 64), dev, 1.0, 2, BF16)
    v = rnd((B, N, 64), dev, 1.0, 3, BF16)
    k[0, 200] = (q[0, 17].float() * 4).to(BF16)       # spike for query 17 in the 4th key tile
    o = torch.empty(B, N, 64, dtype=BF16, device=dev)
    lse = torch.empty(B, H, N, device=dev)
    ops.attn_fwd(q, k, v, o, lse, H, 0.125, False)
    oref, lref = attn_ref(q.float()[:, None], k.float()[:, None], v.float()[:, None], 0.125, False)
    close(o, oref[:, 0], 2e-2, 2e-2, 'spiked attn out')
    close(lse, lref, 1e-3, 2e-3, 'spiked lse')


@pytest.mark.parametrize('Hf,Wf,heads,w,shift', [(14, 14, 3, 7, 0), (14, 14, 3, 7, 3), (8, 16, 2, 4, 2), (7, 7, 4, 7, 0), (16, 16, 2, 8, 4), (21, 14, 1, 7, 3)])
def test_swin_window_attention(dev, Hf, Wf, heads, w, shift):
    from oracle import ref_cpu as R
    from pixparse_amd import ops
    B, hd = 2, 32
    C = heads * hd
    scale = hd ** -0.5
    qkv = rnd((B, Hf, Wf, 3 * C), dev, 1.0, 1, BF16)
    table = rnd(((2 * w - 1) ** 2, heads), dev, 0.5, 2)
    out = torch.empty(B, Hf, Wf, C, dtype=BF16, device=dev)
    ops.swin_attn_fwd(qkv, table, out, B, Hf, Wf, heads, w, shift, scale)
    # oracle: roll -> window partition -> attention with rel-pos bias + shift mask -> reverse -> roll back
    qc = qkv.float().cpu().requires_grad_(True)
    tc = table.cpu().requires_grad_(True)
    h = torch.roll(qc, (-shift, -shift), (1, 2)) if shift else qc
    nW = (Hf // w) * (Wf // w)
    hw = h.view(B, Hf // w, w, Wf // w, w, 3 * C).permute(0, 1, 3, 2, 4, 5).reshape(B * nW, w * w, 3, heads, hd).permute(2, 0, 3, 1, 4)
    bias = tc[R.swin_relative_position_index(w).view(-1)].view(w * w, w * w, heads).permute(2, 0, 1).unsqueeze(0)
    mask = R.swin_shift_mask(Hf, Wf, w, shift)
    if mask is not None:
        bias = bias + mask.repeat(B, 1, 1).unsqueeze(1)
    o = R._attention(hw[0], hw[1], hw[2], scale, False, 'bf16', bias=bias)
    o = o.transpose(1, 2).reshape(B * nW, w * w, C).view(B, Hf // w, Wf // w, w, w, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hf, Wf, C)
    if shift:
        o = torch.roll(o, (shift, shift), (1, 2))
    close(out.cpu(), o.detach(), 2e-2, 2e-2, 'swin attn out')
    d_out = rnd((B, Hf, Wf, C), dev, 1.0, 3, BF16)
    o.backward(d_out.float().cpu())
    dqkv = torch.zeros_like(qkv)
    dtable = torch.zeros_like(table)
    ops.swin_attn_bwd(qkv, table, d_out, dqkv, dtable, B, Hf, Wf, heads, w, shift, scale)
    close(dqkv.cpu(), qc.grad, 3e-2, 3e-2 * float(qc.grad.abs().max()), 'swin dqkv')
    close(dtable.cpu(), tc.grad, 2e-2, 2e-2 * float(tc.grad.abs().max()), 'swin dtable')


# ------------------------------------------------------------------------------------------- loss / optimiser
@pytest.mark.parametrize('M,V', [(37, 515), (300, 1027), (64, 50267)])
def test_cross_entropy(dev, M, V):
    from pixparse_amd import ops
    Vp = (V + 127) // 128 * 128
    logits = torch.zeros(M, Vp, dtype=BF16, device=dev)
    logits[:, :V] = rnd((M, V), dev, 2.0, 1, BF16)
    logits[:, V:] = 55.0                                 # garbage in the padded columns must be ignored
    target = torch.randint(0, V, (M,), device=dev)
    target[::5] = -100
    lf = logits[:, :V].float().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lf, target, ignore_index=-100)
    (ref * 8.0).backward()
    loss = torch.zeros(1, device=dev)
    nv = torch.zeros(1, dtype=torch.int32, device=dev)
    rl = torch.empty(M, device=dev)
    ops.cross_entropy(logits, target, V, 0.5, 8.0, loss, nv, rl, logits)
    assert int(nv) == int((target != -100).sum())
    assert abs(float(loss) - 0.5 * float(ref)) < 1e-5 * abs(float(ref)) + 1e-6
    close(logits[:, :V], lf.grad, 1e-2, 1e-2 * float(lf.grad.abs().max()), 'dlogits')
    assert float(logits[:, V:].abs().max()) == 0.0
    # all targets ignored -> NaN loss like torch, zero gradient
    target[:] = -100
    ops.cross_entropy(logits, target, V, 1.0, 1.0, loss, nv, rl, logits)
    assert math.isnan(float(loss)) and float(logits.abs().max()) == 0.0


def test_grad_norm_adamw(dev, golden_dir):
    import json
    import os
    from safetensors.torch import load_file
    from pixparse_amd import ops
    t = load_file(os.path.join(golden_dir, 'g5_optim.safetensors'))
    meta = json.load(open(os.path.join(golden_dir, 'g5_optim.json')))
    n = meta['n_tensors']
    flat = lambda pre: torch.cat([t[f'{pre}.{i}'].reshape(-1) for i in range(n)]).to(dev)
    p = flat('p0')
    N = p.numel()
    pad = (-N) % 4
    p = torch.cat([p, torch.zeros(pad, device=dev)])
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    pb = torch.empty(p.numel(), dtype=BF16, device=dev)
    state = torch.zeros(4, device=dev)
    for step in range(3):
        g = torch.cat([flat(f'g{step}') * 4.0, torch.zeros(pad, device=dev)])     # pretend loss scale 4
        ops.grad_norm(g, meta['clip'], 0.25, state)
        assert abs(float(state[0]) - meta['grad_norms'][step]) < 1e-5 * meta['grad_norms'][step]
        ops.adamw(p, g, m, v, pb, meta['lr'], meta['betas'][0], meta['betas'][1], meta['eps'], 0.0, step + 1, state, True)
        close(p[:N], flat(f'p{step + 1}'), 1e-6, 1e-7, f'adamw step {step}')
        assert torch.equal(pb, p.to(BF16)) and float(g.abs().max()) == 0.0
    # inf in the grads -> step skipped, flag raised
    g = torch.ones_like(p)
    g[3] = float('inf')
    before = p.clone()
    ops.grad_norm(g, 1.0, 1.0, state)
    ops.adamw(p, g, m, v, pb, 1e-3, 0.9, 0.98, 1e-6, 0.0, 4, state, False)
    assert float(state[2]) == 1.0 and torch.equal(p, before)


def test_adamw_with_device_gradscaler_vs_torch(dev):
    """ArenaAdamW + LossScaler (scale, growth tracker and step counter on the device) against torch.optim.AdamW +
    torch.amp.GradScaler + clip_grad_norm_: six updates with an inf injected into the gradients of update 2 -- the skipped
    step must not advance the bias-correction exponent, the very next update must already run with the halved scale, and
    the scale must grow again after growth_interval clean steps (ref: task/task_cruller_pretrain.py:205-207,259-278)."""
    from pixparse_amd.framework.optim import ArenaAdamW, LossScaler
    from pixparse_amd.layers.arena import ParamArena
    torch.manual_seed(0)
    n = 3 * 4096 + 64
    arena = ParamArena()
    arena.add('w', (n,))
    arena.materialize(dev)
    arena.p.copy_(torch.randn(n, device=dev) * 0.1)
    lr, betas, eps = 1e-2, (0.9, 0.98), 1e-6
    opt = ArenaAdamW(arena, lr=lr, betas=betas, eps=eps, weight_decay=0.0)
    arena.alloc_shadow()
    sc = LossScaler(init_scale=1024.0, growth_interval=2).attach(opt.state)
    pr = torch.nn.Parameter(arena.p.clone())
    topt = torch.optim.AdamW([pr], lr=lr, betas=betas, eps=eps, weight_decay=0.0)
    tsc = torch.amp.GradScaler('cuda', init_scale=1024.0, growth_interval=2)
    grads = [torch.randn(n, device=dev) * (1.0 + i) for i in range(6)]
    grads[1][17] = float('inf')
    want_scales = [1024.0, 512.0, 512.0, 1024.0, 1024.0, 2048.0]      # after each update()
    for i, g in enumerate(grads):
        s_dev = float(opt.state[4])
        assert s_dev == ([1024.0] + want_scales)[i], (i, s_dev)
        arena.g.copy_(g * s_dev)                       # what backward leaves: gradients of the scaled loss
        opt.step(clip_norm=1.0, zero_grad=True, scaler=sc, grad_divisor=1.0)
        pr.grad = g * float(tsc.scale(torch.ones((), device=dev)))     # scaler.scale(loss) initialises / applies the scale
        tsc.unscale_(topt)
        torch.nn.utils.clip_grad_norm_([pr], 1.0)
        tsc.step(topt)
        tsc.update()
        assert float(opt.found_inf()) == (1.0 if i == 1 else 0.0)
        assert sc.get_scale() == want_scales[i] == float(tsc.get_scale()), (i, sc.get_scale(), tsc.get_scale())
        assert opt.step_count == (i + 1 if i < 1 else i), (i, opt.step_count)   # the skipped update does not count
        close(arena.p, pr.detach(), 2e-6, 1e-7, f'update {i}')
        assert torch.equal(arena.pb, arena.p.to(BF16)) and float(arena.g.abs().max()) == 0.0
    assert sc.state_dict()['_growth_tracker'] == 0 and opt.state_dict()['step'] == 5


def test_optimizer_counters_are_exact_beyond_2_pow_24(dev):
    """the step / update counts live as integers on the device (state words 11 / 12): an fp32 counter would stop at 16 777 216 and freeze
    the cosine schedule and the bias corrections; the learning rate of an update far beyond that is the scheduler's closed form"""
    import math
    from pixparse_amd.framework.optim import ArenaAdamW, CosineLRScheduler
    from pixparse_amd.layers.arena import ParamArena
    arena = ParamArena()
    arena.add('w', (256,))
    arena.materialize(dev)
    opt = ArenaAdamW(arena, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0)
    arena.alloc_shadow()
    T = 40_000_000
    sched = CosineLRScheduler(opt, t_initial=T, warmup_t=10, warmup_lr_init=0.0)
    n0 = (1 << 24) + 5
    opt.load_state_dict(dict(step=n0, param_groups=opt.param_groups, exp_avg=arena.m, exp_avg_sq=arena.v))
    opt.set_update_count(n0)
    for i in range(3):
        arena.g.fill_(1e-3)
        opt.step(clip_norm=1.0, zero_grad=True)
        assert opt.step_count == n0 + i + 1, (i, opt.step_count)
        want = 0.5 * 1e-3 * (1 + math.cos(math.pi * (n0 + i) / T))
        assert abs(float(opt.state[8]) - want) < 1e-9, (float(opt.state[8]), want)
    assert int(opt.state[12:13].view(torch.int32).item()) == n0 + 3


def test_adamw_clip_by_value_vs_torch(dev):
    """timm dispatch_clip_grad mode 'value' (ref task_cruller_pretrain.py:270-278) = torch clip_grad_value_ on the UNSCALED gradient,
    applied inside the AdamW launch (state[6]); with a loss scale of 256 on the stored gradients"""
    from pixparse_amd.framework.optim import ArenaAdamW, LossScaler
    from pixparse_amd.layers.arena import ParamArena
    torch.manual_seed(1)
    n = 2 * 4096 + 64
    arena = ParamArena()
    arena.add('w', (n,))
    arena.materialize(dev)
    arena.p.copy_(torch.randn(n, device=dev) * 0.1)
    lr, betas, eps, cv = 1e-2, (0.9, 0.98), 1e-6, 0.3
    opt = ArenaAdamW(arena, lr=lr, betas=betas, eps=eps, weight_decay=0.0)
    arena.alloc_shadow()
    opt.set_clip_value(cv)
    sc = LossScaler(init_scale=256.0, growth_interval=1000).attach(opt.state)
    pr = torch.nn.Parameter(arena.p.clone())
    topt = torch.optim.AdamW([pr], lr=lr, betas=betas, eps=eps, weight_decay=0.0)
    for i in range(3):
        g = torch.randn(n, device=dev) * (0.2 + 0.3 * i)
        assert float((g.abs() > cv).float().mean()) > 0.05       # the clamp is live
        arena.g.copy_(g * 256.0)
        opt.step(clip_norm=None, zero_grad=True, scaler=sc, grad_divisor=1.0)
        pr.grad = g.clone()
        torch.nn.utils.clip_grad_value_([pr], cv)
        topt.step()
        close(arena.p, pr.detach(), 2e-6, 1e-7, f'clip-by-value update {i}')
    opt.set_clip_value(None)
    g = torch.randn(n, device=dev)
    arena.g.copy_(g * 256.0)
    opt.step(clip_norm=None, zero_grad=True, scaler=sc, grad_divisor=1.0)
    pr.grad = g.clone()
    topt.step()
    close(arena.p, pr.detach(), 2e-6, 1e-7, 'clamp off again')


def test_out_of_range_ids_and_targets_do_not_touch_memory(dev):
    """torch raises a device assert for an out-of-vocabulary id / target; here the damage is contained and LOUD:
    embedding rows and the loss turn NaN, the scatter skips the row, nothing outside the tables is read or written"""
    from pixparse_amd import ops
    B, T, D, V = 2, 8, 64, 50
    tok = torch.randn(V, D, device=dev)
    pos = torch.randn(T + 2, D, device=dev)
    ids = torch.randint(0, V, (B, T), device=dev)
    ids[0, 3] = V + 5
    ids[1, 0] = -1
    out = torch.zeros(B * T, D, device=dev)
    ops.embed_fwd(ids, tok, pos, out, 2)
    bad = torch.tensor([3, 8], device=dev)
    assert torch.isnan(out[bad]).all() and torch.isfinite(out[[0, 1, 2, 4, 9]]).all()
    guard = torch.zeros(V + 16, D, device=dev)        # the rows behind the table stay zero
    dpos = torch.zeros(T + 2, D, device=dev)
    dt = torch.ones(B * T, D, device=dev)
    ops.embed_bwd(ids, dt, guard[:V], dpos, 2, False)
    assert float(guard[V:].abs().max()) == 0.0 and float(guard[:V].sum()) == (B * T - 2) * D
    M, Vp = 6, 64
    logits = torch.randn(M, Vp, device=dev).to(BF16)
    target = torch.tensor([1, 2, Vp + 100, -100, 3, 49], device=dev)
    loss = torch.zeros(1, device=dev); nv = torch.zeros(1, dtype=torch.int32, device=dev); rl = torch.empty(M, device=dev)
    ops.cross_entropy(logits, target, V, 1.0, 1.0, loss, nv, rl, logits)
    assert math.isnan(float(loss)) and float(logits[2].abs().max()) == 0.0 and torch.isfinite(logits.float()).all()


def test_cross_entropy_device_side_scale(dev):
    """grad_mul_dev: the loss scale is read on the device (GradScaler word state[4])"""
    from pixparse_amd import ops
    M, V, Vp = 16, 100, 128
    logits = (torch.randn(M, Vp, device=dev) * 2).to(BF16)
    target = torch.randint(0, V, (M,), device=dev)
    a, b = logits.clone(), logits.clone()
    loss = torch.zeros(1, device=dev); nv = torch.zeros(1, dtype=torch.int32, device=dev); rl = torch.empty(M, device=dev)
    ops.cross_entropy(a, target, V, 1.0, 0.5 * 256.0, loss, nv, rl, a)
    la = float(loss)
    scale = torch.tensor([256.0], device=dev)
    ops.cross_entropy(b, target, V, 1.0, 0.5, loss, nv, rl, b, scale)
    assert torch.equal(a, b) and float(loss) == la


@pytest.mark.parametrize('H,W,C,Ho,Wo', [(877, 620, 3, 1280, 960), (1754, 1240, 1, 960, 640), (300, 200, 3, 224, 224)])
def test_gpu_image_preprocess(dev, H, W, C, Ho, Wo):
    """uint8 page -> ToTensor -> bicubic antialias Resize -> Normalize on the GPU vs the same ops in torch (CPU)"""
    import torch.nn.functional as F
    from pixparse_amd.data.gpu_preprocess import GpuImagePreprocess
    g = torch.Generator().manual_seed(1)
    img = torch.randint(0, 256, (H, W, C), generator=g, dtype=torch.uint8)
    mean, std = [0.48, 0.45, 0.40][:C], [0.27, 0.26, 0.28][:C]
    pre = GpuImagePreprocess((Ho, Wo), mean, std, C, dev)
    out = pre(img)
    x = img.permute(2, 0, 1).float() / 255.0
    ref = F.interpolate(x[None], size=(Ho, Wo), mode='bicubic', antialias=True, align_corners=False)[0]
    ref = (ref - torch.tensor(mean).view(-1, 1, 1)) / torch.tensor(std).view(-1, 1, 1)
    assert out.shape == (C, Ho, Wo)
    assert float((out.cpu() - ref).abs().max()) < 2e-5


# ------------------------------------------------------------------------------------------- dropout
def test_dropout_kernels(dev):
    """crl_dropout / crl_dropout_add / crl_dropout_mask: one stateless Philox mask per (seed, step, site) shared by all three, keep
    rate 1 - p, exact arithmetic given the mask, different sites / steps / seeds decorrelated, p = 0 is the identity"""
    from pixparse_amd import ops
    n = 1 << 20
    d = ops.DropSpec(0.1, 1234, 7)
    keep = ops.dropout_mask(n, d, 3, dev).bool()
    rate = float(keep.float().mean())
    assert abs(rate - 0.9) < 3 * math.sqrt(0.09 / n) + 2e-5, rate            # binomial 3 sigma (+ the 16-bit threshold granularity)
    assert torch.equal(keep, ops.dropout_mask(n, d, 3, dev).bool())           # pure function of its arguments
    for other in (ops.dropout_mask(n, d, 4, dev), ops.dropout_mask(n, ops.DropSpec(0.1, 1234, 8), 3, dev), ops.dropout_mask(n, ops.DropSpec(0.1, 1235, 7), 3, dev)):
        agree = float((other.bool() == keep).float().mean())
        assert abs(agree - 0.82) < 5e-3, agree                                # independent masks agree with probability .81 + .01
    x = rnd((n,), dev, 1.0, 1, BF16)
    y = torch.empty_like(x)
    ops.dropout(x, y, d, 3)
    want = torch.where(keep, (x.float() * (1.0 / 0.9)).to(BF16), torch.zeros_like(x))
    assert torch.equal(y, want)
    xin = x.clone()
    ops.dropout(xin, xin, d, 3)                                               # in place
    assert torch.equal(xin, want)
    f = rnd((n,), dev, 1.0, 2)
    g, gb = torch.empty_like(f), torch.empty(n, dtype=BF16, device=dev)
    ops.dropout(f, g, d, 3, y_bf16=gb)
    wantf = torch.where(keep, f * (1.0 / 0.9), torch.zeros_like(f))
    assert torch.allclose(g, wantf, rtol=1e-6, atol=0) and torch.equal(gb, g.to(BF16))
    r = rnd((n,), dev, 1.0, 3)
    out = torch.empty_like(r)
    ops.dropout_add(x, r, out, d, 3)
    assert torch.equal(out, r + want.float())
    ops.dropout(x, y, ops.DropSpec(0.0, 1, 1), 0)
    assert torch.equal(y, x)
