"""Same-box yardstick next to `roofline.peak_measured`: the vendor library's bf16 GEMM (torch.matmul -> hipBLASLt) and
torch's SDPA forward + backward on the shapes of the cfg-3 step, alternating with this library's kernels on ONE device, random data,
back-to-back launches.  scripts/ only: nothing here is imported by pixparse_amd/ or bench.py.

  python scripts/yardstick.py            # the table (TF/s per row, N launches back to back, R alternating rounds)
  python scripts/yardstick.py pmc        # few launches of every row: run under `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace` for the clocks
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pixparse_amd import hip, ops
dev = torch.device('cuda:0'); BF16 = torch.bfloat16
PMC = len(sys.argv) > 1 and sys.argv[1] == 'pmc'
NL = 6 if PMC else int(os.environ.get('YARD_LAUNCHES', 200))
ROUNDS = 1 if PMC else 3


def timed(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def gemm_rows():
    shapes = [('square 8192', 8192, 8192, 8192), ('fc1 49512x4096x1024', 49512, 4096, 1024), ('proj 49512x1024x1024', 49512, 1024, 1024),
              ('fc2 49512x1024x4096', 49512, 1024, 4096), ('qkv 49512x3072x1024', 49512, 3072, 1024)]
    for name, M, N, K in shapes:
        x = torch.randn(M, K, device=dev).to(BF16); w = torch.randn(N, K, device=dev).to(BF16)
        out = torch.empty(M, N, dtype=BF16, device=dev); out2 = torch.empty(M, N, dtype=BF16, device=dev)
        wt = w.t()
        f_vendor = lambda: torch.matmul(x, wt, out=out2)
        f_ours = lambda: ops.linear_fwd(x, w, None, out)
        n = max(20, int(NL * min(1.0, 1.1e12 / (2.0 * M * N * K)) + 0.5)) if not PMC else NL
        for r in range(ROUNDS):
            for who, f in (('hipBLASLt (torch.matmul)', f_vendor), ('crl_gemm_bf16', f_ours)):
                ms = timed(f, n)
                print(f'{name:22s} {who:26s} round {r}: {n:4d} launches {ms * 1000:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TF/s', flush=True)
        err = (out.float() - out2.float()).abs().max().item() / out2.float().abs().max().item()
        print(f'{name:22s} max |ours - vendor| / max |vendor| = {err:.2e}', flush=True)
        del x, w, out, out2


def attn_rows():
    import torch.nn.functional as F
    B, H, N, d = 8, 16, 6189, 64
    D = H * d
    qkv = (torch.randn(B, N, 3 * D, device=dev) * 0.5).to(BF16)
    o = torch.empty(B, N, D, dtype=BF16, device=dev); lse = torch.empty(B, H, N, device=dev)
    do = torch.randn(B, N, D, device=dev).to(BF16); dqkv = torch.empty_like(qkv); delta = torch.empty(2, B, H, N, device=dev)
    q, k, v = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
    qp = (q.float() * (0.125 * ops.LOG2E)).to(BF16)
    fl = 4.0 * N * N * D * B
    # vendor: [B, H, N, d] views of the same projections (no copies: SDPA takes strided inputs)
    qh = q.reshape(B, N, H, d).transpose(1, 2).detach().requires_grad_(True)
    kh = k.reshape(B, N, H, d).transpose(1, 2).detach().requires_grad_(True)
    vh = v.reshape(B, N, H, d).transpose(1, 2).detach().requires_grad_(True)
    doh = do.reshape(B, N, H, d).transpose(1, 2)

    def vendor_fwd():
        with torch.no_grad():
            return F.scaled_dot_product_attention(qh, kh, vh)

    def vendor_fwd_bwd():
        out = F.scaled_dot_product_attention(qh, kh, vh)
        out.backward(doh)
        qh.grad = kh.grad = vh.grad = None

    ours_fwd = lambda: ops.attn_fwd(qp, k, v, o, lse, H, 0.125, False, q_prescaled=True)

    def ours_fwd_bwd():
        ops.attn_fwd(qp, k, v, o, lse, H, 0.125, False, q_prescaled=True)
        ops.attn_bwd(qp, k, v, o, do, lse, delta, dqkv[:, :, :D], dqkv[:, :, D:2 * D], dqkv[:, :, 2 * D:], H, 0.125, False, q_prescaled=True)

    n = 4 if PMC else 40
    for r in range(ROUNDS):
        for who, f, mult in (('torch SDPA fwd', vendor_fwd, 1), ('crl_attn_fwd (prescaled q)', ours_fwd, 1),
                             ('torch SDPA fwd+bwd', vendor_fwd_bwd, 3), ('crl_attn_fwd + crl_attn_bwd', ours_fwd_bwd, 3)):
            try:
                ms = timed(f, n)
                print(f'attention B8 H16 N6189 {who:28s} round {r}: {n:3d} x {ms:8.3f} ms  {mult * fl / ms / 1e9:7.1f} TF/s algorithmic', flush=True)
            except Exception as e:      # a vendor path that is missing is a result too
                print(f'attention B8 H16 N6189 {who:28s} FAILED: {type(e).__name__}: {str(e)[:200]}', flush=True)


if __name__ == '__main__':
    print('device:', torch.cuda.get_device_name(0), '| torch', torch.__version__, '| launches per row', NL, flush=True)
    hip.call('crl_gemm_set_policy', 2)      # the 256x256 kernel on every shape: kernel against kernel (the automatic plan adds the remainder split)
    gemm_rows()
    hip.call('crl_gemm_set_policy', 0)
    attn_rows()
